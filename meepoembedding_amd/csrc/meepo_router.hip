// meepo_router.hip — hashing KATs on device, stable shard partition and row (un)permutation for gfx950.
//
// Reference anchor: /root/reference/README.md:2 ("A distributed … Embedding"); no code upstream.  Semantics:
// SPEC.md §1 (hashing) and §5 (sharding).  The partition is a three-kernel counting sort by owner that keeps
// batch order inside each shard segment: per-block histograms (wave ballot + popcount per shard), a per-shard
// scan over blocks (one wave per shard, shuffle prefix-sum), and a scatter that recomputes each key's rank
// from the same ballots.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "meepo_device.h"
#include "meepo_host.h"

struct mee_router {
    int device;
    uint64_t max_batch;
    uint32_t n_shards;
    uint32_t max_blocks;
    uint32_t* blockcnt;  // [max_blocks][n_shards] counts, then exclusive offsets inside the shard segment
    uint64_t* base;      // [n_shards] start of each shard segment
};

struct mee_p2p {
    int device;
    uint32_t n_shards, rank, dim;
    uint64_t cap, max_batch;
    // local symmetric buffers, exported to the peers
    int64_t* inbox_keys;   // [n_shards][cap]
    int32_t* inbox_dst;    // [n_shards][cap]
    uint32_t* inbox_cnt;   // [n_shards] fill counts, followed by [n_shards] barrier flags (flag q = the last epoch rank q announced here)
    float* out;            // [max_batch][dim]
    uint8_t* found;        // [max_batch]
    float* inbox_rows;     // [n_shards][cap][dim] payload rows of pushed (key, row) pairs; null unless created with payload
    uint32_t* status;      // [0] bit 0: a segment overflowed `cap`, bit 1: a barrier timed out; [1] barriers executed so far (the epoch)
    // device-resident pointer tables (index = rank) and their host copies
    void** d_tables;       // kP2PBuffers tables of n_shards pointers each
    void* h_tables[6][64];
    bool connected;
};

namespace mee {

constexpr int kPartBlock = 1024;  // keys per block in the partition kernels (one per thread): 4x fewer rows for the scan
constexpr int kMaxShards = 64;

__global__ void hash_batch_kernel(const int64_t* __restrict__ keys, uint64_t n, uint64_t nb, uint32_t g, uint64_t* mix_out,
                                  uint64_t* bucket_out, uint32_t* owner_out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t k = keys[i];
        if (mix_out) mix_out[i] = mix64((uint64_t)k);
        if (bucket_out) bucket_out[i] = bucket_of(k, nb);
        if (owner_out) owner_out[i] = owner_of(k, g);
    }
}

// SKIP_PAD: EMPTY keys (padding, SPEC.md §2) belong to nobody: not counted, not sent
template <bool SKIP_PAD>
__global__ __launch_bounds__(kPartBlock) void part_count_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t g,
                                                                uint32_t* blockcnt) {
    __shared__ uint32_t wcnt[kPartBlock / 64][kMaxShards];
    const uint32_t i = blockIdx.x * kPartBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t key = i < n ? keys[i] : kEmpty;
    const bool inb = i < n && !(SKIP_PAD && key == kEmpty);
    const uint32_t o = inb ? owner_of(key, g) : 0xFFFFFFFFu;
    for (uint32_t p = 0; p < g; ++p) {
        const uint64_t m = __ballot(o == p);
        if (lane == 0) wcnt[w][p] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < g) {
        uint32_t c = 0;
#pragma unroll
        for (int ww = 0; ww < kPartBlock / 64; ++ww) c += wcnt[ww][threadIdx.x];
        blockcnt[(uint64_t)blockIdx.x * g + threadIdx.x] = c;
    }
}

// one block; wave w scans shards w, w+nwaves, … over all key blocks
__global__ __launch_bounds__(1024) void part_scan_kernel(uint32_t* blockcnt, uint32_t n_blocks, uint32_t g, uint64_t* base,
                                                         uint64_t* counts_out) {
    __shared__ uint64_t total[kMaxShards];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (uint32_t p = w; p < g; p += nw) {
        uint32_t run = 0;
        for (uint32_t b0 = 0; b0 < n_blocks; b0 += 64) {
            const uint32_t b = b0 + lane;
            const uint32_t c = b < n_blocks ? blockcnt[(uint64_t)b * g + p] : 0;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (b < n_blocks) blockcnt[(uint64_t)b * g + p] = run + incl - c;
            run += __shfl(incl, 63);
        }
        if (lane == 0) total[p] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t p = 0; p < g; ++p) {
            base[p] = acc;
            counts_out[p] = total[p];
            acc += total[p];
        }
    }
}

template <bool SKIP_PAD>
__global__ __launch_bounds__(kPartBlock) void part_scatter_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t g,
                                                                  const uint32_t* __restrict__ blockoff,
                                                                  const uint64_t* __restrict__ base, int64_t* send_keys,
                                                                  int64_t* perm) {
    __shared__ uint32_t wcnt[kPartBlock / 64][kMaxShards];
    const uint32_t i = blockIdx.x * kPartBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t key = i < n ? keys[i] : kEmpty;
    const bool inb = i < n && !(SKIP_PAD && key == kEmpty);
    const uint32_t o = inb ? owner_of(key, g) : 0xFFFFFFFFu;
    uint32_t r = 0;
    for (uint32_t p = 0; p < g; ++p) {
        const uint64_t m = __ballot(o == p);
        if (o == p) r = (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wcnt[w][p] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (inb) {
        for (int ww = 0; ww < w; ++ww) r += wcnt[ww][o];
        const uint64_t dst = base[o] + blockoff[(uint64_t)blockIdx.x * g + o] + r;
        send_keys[dst] = key;
        perm[dst] = (int64_t)i;
    }
}

// SCATTER: out[perm[q]] = rows[q];  else out[q] = rows[perm[q]].  One element of type T per thread step.
template <typename T, bool SCATTER>
__global__ void permute_rows_kernel(const T* __restrict__ rows, const int64_t* __restrict__ perm, uint64_t n, uint32_t epr,
                                    T* __restrict__ out) {
    const uint64_t total = n * epr;
    for (uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = idx / epr;
        const uint32_t e = (uint32_t)(idx - q * epr);
        const uint64_t p = (uint64_t)perm[q];
        if (SCATTER) out[p * epr + e] = rows[idx];
        else out[idx] = rows[p * epr + e];
    }
}

template <bool SCATTER>
static int permute_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream,
                        const char* name) {
    if (n && (!d_rows || !d_perm || !d_out)) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (n == 0 || row_bytes == 0) return MEE_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)d_rows % 16 == 0) && ((uintptr_t)d_out % 16 == 0);
    const bool a4 = row_bytes % 4 == 0 && ((uintptr_t)d_rows % 4 == 0) && ((uintptr_t)d_out % 4 == 0);
    if (a16) {
        const uint32_t epr = (uint32_t)(row_bytes / 16);
        permute_rows_kernel<float4, SCATTER><<<grid_for(n * epr, 256, 1u << 16), 256, 0, st>>>((const float4*)d_rows, d_perm, n, epr, (float4*)d_out);
    } else if (a4) {
        const uint32_t epr = (uint32_t)(row_bytes / 4);
        permute_rows_kernel<uint32_t, SCATTER><<<grid_for(n * epr, 256, 1u << 16), 256, 0, st>>>((const uint32_t*)d_rows, d_perm, n, epr, (uint32_t*)d_out);
    } else {
        const uint32_t epr = (uint32_t)row_bytes;
        permute_rows_kernel<uint8_t, SCATTER><<<grid_for(n * epr, 256, 1u << 16), 256, 0, st>>>((const uint8_t*)d_rows, d_perm, n, epr, (uint8_t*)d_out);
    }
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// ---- peer-to-peer sharded find (SPEC.md §5 without the all-to-alls) ---------------------------------------------
// Every rank owns five "symmetric" buffers that its peers map through HIP IPC: an inbox of keys and of destination
// indices with one segment per source rank, the segment fill counts, and its result rows / found bytes.  A lookup is
//   push:  each rank's partitioned keys (+ their batch positions) are stored straight into the owners' inboxes (xGMI)
//   find:  each owner probes what arrived and stores every row straight into the requester's result buffer (xGMI)
// with one stream-ordered barrier after each phase.  No all-to-all, no un-permute pass, no host sync.
constexpr int kP2PBuffers = 6;  // inbox keys, inbox dst, inbox counts, result rows, result found bytes, inbox payload rows
struct P2PPeers {            // device-resident pointer tables, index = rank
    int64_t** keys; int32_t** dst; uint32_t** cnt; float** out; uint8_t** found; float** rows;
};

// grid: x over positions inside a segment, y = destination rank
__global__ __launch_bounds__(256) void p2p_push_kernel(const int64_t* __restrict__ send_keys, const int64_t* __restrict__ perm,
                                                       const uint64_t* __restrict__ counts, const uint64_t* __restrict__ base,
                                                       P2PPeers peers, uint32_t me, uint64_t cap, uint32_t* status, int pad) {
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p];
    const uint64_t take = cnt < cap ? cnt : cap;
    int64_t* __restrict__ dk = peers.keys[p] + (uint64_t)me * cap;
    int32_t* __restrict__ dd = peers.dst[p] + (uint64_t)me * cap;
    const uint64_t b0 = base[p];
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < take; j += (uint64_t)gridDim.x * blockDim.x) {
        dk[j] = send_keys[b0 + j];
        dd[j] = (int32_t)perm[b0 + j];
    }
    if (pad)  // the owner will hand the whole fixed-size inbox to an operator: unused positions become padding keys
        for (uint64_t j = take + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < cap; j += (uint64_t)gridDim.x * blockDim.x) dk[j] = kEmpty;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        peers.cnt[p][me] = (uint32_t)take;
        if (cnt > cap) atomicOr(status, 1u);
    }
}

// Cross-rank barrier over the peer-mapped flag words: every rank announces epoch e in the flag word it owns on every peer
// (system-scope release store: the rank's earlier kernels — whose stores into peer memory were made visible by their kernel
// end — are ordered before it), then waits until all peers have announced e here.  One wave; the wait is bounded (wall clock)
// so the grid always drains: on time-out bit 1 of the status word is set and the caller must not trust the step.
__global__ __launch_bounds__(64) void p2p_barrier_kernel(P2PPeers peers, uint32_t me, uint32_t n_shards, uint32_t* status,
                                                         unsigned long long timeout_ticks) {
    const uint32_t q = threadIdx.x;
    // the epoch lives in device memory and is advanced by the kernel itself (every rank runs the same number of barriers),
    // so a captured launch stays correct when it is replayed
    uint32_t epoch = 0;
    if (q == 0) { epoch = status[1] + 1; status[1] = epoch; }
    epoch = __shfl(epoch, 0);
    if (q < n_shards) {
        uint32_t* theirs = peers.cnt[q] + n_shards + me;   // my flag word at rank q
        __hip_atomic_store(theirs, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* mine = peers.cnt[me] + n_shards + q;   // rank q's flag word here
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            if (wall_clock64() - t0 > timeout_ticks) { atomicOr(status, 2u); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// payload rows of a push: one 16-lane tile copies row perm[b0+j] of the batch into position j of the owner's segment
__global__ __launch_bounds__(256) void p2p_push_rows_kernel(const float4* __restrict__ rows, const int64_t* __restrict__ perm,
                                                            const uint64_t* __restrict__ counts, const uint64_t* __restrict__ base,
                                                            P2PPeers peers, uint32_t me, uint64_t cap, uint32_t dim4) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p];
    const uint64_t take = cnt < cap ? cnt : cap;
    float4* __restrict__ dr = reinterpret_cast<float4*>(peers.rows[p]) + (uint64_t)me * cap * dim4;
    const uint64_t b0 = base[p];
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t j0 = wave * 4; j0 < take; j0 += n_waves * 4) {
        const uint64_t j = j0 + tile;
        if (j >= take) continue;
        const uint64_t src = (uint64_t)perm[b0 + j] * dim4;
        for (uint32_t c = tl; c < dim4; c += 16) dr[j * dim4 + c] = rows[src + c];
    }
}

// grid: x over the positions of one inbox segment, y = source rank.  Same shape as find_kernel: R keys in flight per
// 16-lane tile (bucket lines requested back to back, then the rows); DIM4 = dim/4 when 16 or 32, else 0 (run-time dim).
// Rows and found bytes are stored into the requester's buffers — peer memory, i.e. xGMI stores, for remote sources.
template <int DIM4, int R>
__global__ __launch_bounds__(256) void p2p_find_kernel(const int64_t* __restrict__ tkeys, const float4* __restrict__ values, uint64_t nb,
                                                       uint32_t dim4_rt, float defv, const int64_t* __restrict__ in_keys,
                                                       const int32_t* __restrict__ in_dst, const uint32_t* __restrict__ in_cnt,
                                                       P2PPeers peers, uint64_t cap) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t s = blockIdx.y;
    const uint32_t cnt = in_cnt[s];
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const int64_t* __restrict__ seg_keys = in_keys + (uint64_t)s * cap;
    const int32_t* __restrict__ seg_dst = in_dst + (uint64_t)s * cap;
    float4* __restrict__ outp = reinterpret_cast<float4*>(peers.out[s]);
    uint8_t* __restrict__ fnd = peers.found[s];
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    const float4 def4 = make_float4(defv, defv, defv, defv);
    for (uint32_t base = wave * 4 * R; base < cnt; base += n_waves * 4 * R) {  // wave-uniform
        int64_t key[R], slot[R];
        int32_t dst[R];
        bool inb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + r * 4 + tile;
            inb[r] = i < cnt;
            key[r] = inb[r] ? seg_keys[i] : kEmpty;
            dst[r] = inb[r] ? seg_dst[i] : 0;
        }
        uint64_t b[R];
        int64_t kb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            b[r] = bucket_of(key[r], nb);
            kb[r] = (inb[r] && !reserved_key(key[r])) ? tkeys[b[r] * kW + tl] : kEmpty;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            slot[r] = -1;
            bool pend = inb[r] && !reserved_key(key[r]);
            uint64_t bb = b[r], steps = 0;
            int64_t k = kb[r];
            while (true) {
                const uint32_t tm = tile_bits(__ballot(pend && k == key[r]), tile);
                const uint32_t te = tile_bits(__ballot(pend && k == kEmpty), tile);
                if (pend) {
                    if (tm) { slot[r] = (int64_t)(bb * kW) + (__ffs(tm) - 1); pend = false; }
                    else if (te || ++steps >= nb) pend = false;
                    else bb = next_bucket(bb, step_of(key[r], nb), nb);
                }
                if (!__any(pend)) break;
                k = pend ? tkeys[bb * kW + tl] : kEmpty;
            }
        }
        if constexpr (DIM4 != 0) {
            constexpr int C = DIM4 / 16;
            float4 row[R][C];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c) row[r][c] = slot[r] >= 0 ? values[(uint64_t)slot[r] * DIM4 + c * 16 + tl] : def4;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (inb[r]) {
#pragma unroll
                    for (int c = 0; c < C; ++c) outp[(uint64_t)dst[r] * DIM4 + c * 16 + tl] = row[r][c];
                }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (inb[r])
                    for (uint32_t c = tl; c < dim4; c += 16)
                        outp[(uint64_t)dst[r] * dim4 + c] = slot[r] >= 0 ? values[(uint64_t)slot[r] * dim4 + c] : def4;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (inb[r] && tl == 0) fnd[dst[r]] = slot[r] >= 0;
    }
}

}  // namespace mee

using namespace mee;

extern "C" {

int mee_hash_batch(const int64_t* d_keys, size_t n, uint64_t n_buckets, uint32_t n_shards, uint64_t* d_mix_out,
                   uint64_t* d_bucket_out, uint32_t* d_owner_out, void* stream) {
    MEE_RANGE("mee_hash_batch");
    if (n && !d_keys) return fail(MEE_ERR_INVALID_ARG, "mee_hash_batch: null keys");
    if (n == 0) return MEE_OK;
    hash_batch_kernel<<<grid_for(n, 256, 1u << 14), 256, 0, (hipStream_t)stream>>>(d_keys, n, n_buckets, n_shards, d_mix_out,
                                                                                 d_bucket_out, d_owner_out);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_router_destroy(mee_router* r) {
    MEE_RANGE("mee_router_destroy");
    if (!r) return MEE_OK;
    DeviceGuard g(r->device);
    if (r->blockcnt) (void)hipFree(r->blockcnt);
    if (r->base) (void)hipFree(r->base);
    delete r;
    return MEE_OK;
}

int mee_router_create(int32_t device, uint64_t max_batch, uint32_t n_shards, mee_router** out) {
    MEE_RANGE("mee_router_create");
    if (!out) return fail(MEE_ERR_INVALID_ARG, "mee_router_create: null out");
    *out = nullptr;
    if (n_shards == 0 || n_shards > (uint32_t)kMaxShards || max_batch == 0 || max_batch > (1ull << 30))
        return fail(MEE_ERR_INVALID_ARG, "mee_router_create: n_shards must be 1..%d and max_batch 1..2^30", kMaxShards);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MEE_ERR_NO_DEVICE, "mee_router_create: no HIP device visible (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MEE_ERR_INVALID_ARG, "mee_router_create: device %d out of range", device);
    DeviceGuard g(device);
    mee_router* r = new (std::nothrow) mee_router();
    if (!r) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    r->device = device; r->max_batch = max_batch; r->n_shards = n_shards;
    r->max_blocks = (uint32_t)((max_batch + kPartBlock - 1) / kPartBlock);
    r->blockcnt = nullptr; r->base = nullptr;
    if (hipMalloc((void**)&r->blockcnt, (size_t)r->max_blocks * n_shards * 4) != hipSuccess ||
        hipMalloc((void**)&r->base, n_shards * 8) != hipSuccess) {
        mee_router_destroy(r);
        return fail(MEE_ERR_OUT_OF_MEMORY, "mee_router_create: hipMalloc failed");
    }
    *out = r;
    return MEE_OK;
}

static int partition_common(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm,
                            void* stream, bool skip_pad, const char* name) {
    if (!r || !d_counts || (n && (!d_keys || !d_send_keys || !d_perm))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (n > r->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: n=%zu exceeds max_batch=%llu", name, n, (unsigned long long)r->max_batch);
    DeviceGuard g(r->device);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nblk = (uint32_t)((n + kPartBlock - 1) / kPartBlock);
    if (nblk) {
        if (skip_pad) part_count_kernel<true><<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt);
        else part_count_kernel<false><<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt);
    }
    part_scan_kernel<<<1, 1024, 0, st>>>(r->blockcnt, nblk, r->n_shards, r->base, d_counts);
    if (nblk) {
        if (skip_pad) part_scatter_kernel<true><<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt, r->base, d_send_keys, d_perm);
        else part_scatter_kernel<false><<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt, r->base, d_send_keys, d_perm);
    }
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}
int mee_partition(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm,
                  void* stream) {
    MEE_RANGE("mee_partition");
    return partition_common(r, d_keys, n, d_send_keys, d_counts, d_perm, stream, false, "mee_partition");
}
int mee_partition_padded(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm,
                         void* stream) {
    MEE_RANGE("mee_partition_padded");
    return partition_common(r, d_keys, n, d_send_keys, d_counts, d_perm, stream, true, "mee_partition_padded");
}

// ---- peer-to-peer exchange: lifetime and IPC ----------------------------------------------------------------------
int mee_p2p_destroy(mee_p2p* c) {
    MEE_RANGE("mee_p2p_destroy");
    if (!c) return MEE_OK;
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    if (c->connected)
        for (int b = 0; b < kP2PBuffers; ++b)
            for (uint32_t p = 0; p < c->n_shards; ++p)
                if (p != c->rank && c->h_tables[b][p]) (void)hipIpcCloseMemHandle(c->h_tables[b][p]);
    void* mine[] = {c->inbox_keys, c->inbox_dst, c->inbox_cnt, c->out, c->found, c->inbox_rows, c->status, c->d_tables};
    for (void* p : mine) if (p) (void)hipFree(p);
    delete c;
    return MEE_OK;
}

int mee_p2p_create(int32_t device, uint32_t n_shards, uint32_t rank, uint64_t slots_per_peer, uint64_t max_batch, uint32_t dim,
                   int with_payload, mee_p2p** out) {
    MEE_RANGE("mee_p2p_create");
    if (!out) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_create: null out");
    *out = nullptr;
    if (n_shards == 0 || n_shards > (uint32_t)kMaxShards || rank >= n_shards || slots_per_peer == 0 || max_batch == 0 ||
        max_batch > (1ull << 30) || dim < 4 || (dim & 3))
        return fail(MEE_ERR_INVALID_ARG, "mee_p2p_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MEE_ERR_NO_DEVICE, "mee_p2p_create: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_create: device %d out of range", device);
    DeviceGuard g(device);
    mee_p2p* c = new (std::nothrow) mee_p2p();
    if (!c) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    memset(c, 0, sizeof *c);
    c->device = device; c->n_shards = n_shards; c->rank = rank; c->dim = dim; c->cap = slots_per_peer; c->max_batch = max_batch;
    const uint64_t slots = (uint64_t)n_shards * slots_per_peer;
    // The five peer-visible buffers are FINE-GRAINED device memory: peers store into them from inside kernels and the
    // owner reads them in its next kernel with nothing but a collective in between, so they must stay coherent across
    // GPUs without relying on a system-scope cache invalidate at the reader's kernel start (coarse-grained hipMalloc
    // memory is only guaranteed coherent at host-visible synchronisation points).  Set MEE_P2P_COARSE=1 to use plain
    // hipMalloc instead (single-GPU rehearsals).
    const bool fine = getenv("MEE_P2P_COARSE") == nullptr;
    auto sym_alloc = [&](void** p, size_t bytes) {
        return fine ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) : hipMalloc(p, bytes);
    };
    if (sym_alloc((void**)&c->inbox_keys, slots * 8) != hipSuccess || sym_alloc((void**)&c->inbox_dst, slots * 4) != hipSuccess ||
        sym_alloc((void**)&c->inbox_cnt, 2 * n_shards * 4) != hipSuccess || sym_alloc((void**)&c->out, (max_batch + 1) * (uint64_t)dim * 4) != hipSuccess ||   // + one spare row: see mee_p2p_buffers
        sym_alloc((void**)&c->found, max_batch + 1) != hipSuccess ||
        (with_payload && sym_alloc((void**)&c->inbox_rows, slots * (uint64_t)dim * 4) != hipSuccess) ||
        hipMalloc((void**)&c->status, 8) != hipSuccess ||
        hipMalloc((void**)&c->d_tables, kP2PBuffers * (size_t)n_shards * sizeof(void*)) != hipSuccess) {
        mee_p2p_destroy(c);
        return fail(MEE_ERR_OUT_OF_MEMORY, "mee_p2p_create: device allocation failed (%s)", fine ? "fine-grained" : "coarse-grained");
    }
    if (hipMemset(c->inbox_cnt, 0, 2 * n_shards * 4) != hipSuccess || hipMemset(c->status, 0, 8) != hipSuccess) {
        mee_p2p_destroy(c);
        return fail(MEE_ERR_HIP, "mee_p2p_create: hipMemset failed");
    }
    *out = c;
    return MEE_OK;
}

static void* p2p_local(const mee_p2p* c, int b) {
    switch (b) { case 0: return c->inbox_keys; case 1: return c->inbox_dst; case 2: return c->inbox_cnt; case 3: return c->out; case 4: return c->found; default: return c->inbox_rows; }
}

int mee_p2p_export(mee_p2p* c, void* handles) {
    MEE_RANGE("mee_p2p_export");
    if (!c || !handles) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == MEE_IPC_HANDLE_BYTES, "IPC handle size");
    DeviceGuard g(c->device);
    memset(handles, 0, (size_t)kP2PBuffers * MEE_IPC_HANDLE_BYTES);
    for (int b = 0; b < kP2PBuffers; ++b) {
        if (!p2p_local(c, b)) continue;  // no payload inbox
        hipIpcMemHandle_t h;
        MEE_HIP(hipIpcGetMemHandle(&h, p2p_local(c, b)));
        memcpy((char*)handles + b * MEE_IPC_HANDLE_BYTES, &h, MEE_IPC_HANDLE_BYTES);
    }
    return MEE_OK;
}

int mee_p2p_connect(mee_p2p* c, const void* all_handles) {
    MEE_RANGE("mee_p2p_connect");
    if (!c || !all_handles) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_connect: null argument");
    if (c->connected) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_connect: already connected");
    DeviceGuard g(c->device);
    for (uint32_t p = 0; p < c->n_shards; ++p)
        for (int b = 0; b < kP2PBuffers; ++b) {
            if (p == c->rank || !p2p_local(c, b)) { c->h_tables[b][p] = p2p_local(c, b); continue; }
            hipIpcMemHandle_t h;
            memcpy(&h, (const char*)all_handles + ((size_t)p * kP2PBuffers + b) * MEE_IPC_HANDLE_BYTES, MEE_IPC_HANDLE_BYTES);
            void* ptr = nullptr;
            MEE_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
            c->h_tables[b][p] = ptr;
        }
    for (int b = 0; b < kP2PBuffers; ++b)
        MEE_HIP(hipMemcpy(c->d_tables + (size_t)b * c->n_shards, c->h_tables[b], c->n_shards * sizeof(void*), hipMemcpyHostToDevice));
    c->connected = true;
    return MEE_OK;
}

int mee_p2p_buffers(mee_p2p* c, float** d_out, uint8_t** d_found) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_buffers: null argument");
    if (d_out) *d_out = c->out;
    if (d_found) *d_found = c->found;
    return MEE_OK;
}

static P2PPeers p2p_peers(const mee_p2p* c) {
    P2PPeers pp;
    pp.keys = (int64_t**)(c->d_tables + 0 * (size_t)c->n_shards);
    pp.dst = (int32_t**)(c->d_tables + 1 * (size_t)c->n_shards);
    pp.cnt = (uint32_t**)(c->d_tables + 2 * (size_t)c->n_shards);
    pp.out = (float**)(c->d_tables + 3 * (size_t)c->n_shards);
    pp.found = (uint8_t**)(c->d_tables + 4 * (size_t)c->n_shards);
    pp.rows = (float**)(c->d_tables + 5 * (size_t)c->n_shards);
    return pp;
}

int mee_p2p_push(mee_p2p* c, mee_router* r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts, size_t n,
                 void* stream) {
    MEE_RANGE("mee_p2p_push");
    if (!c || !r || !d_counts || (n && (!d_send_keys || !d_perm))) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push: null argument");
    if (!c->connected) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push: not connected");
    if (r->n_shards != c->n_shards || n > c->max_batch) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push: router/batch mismatch");
    DeviceGuard g(c->device);
    const dim3 grid(grid_for(n / c->n_shards + 256, 256, 1024), c->n_shards);
    p2p_push_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_send_keys, d_perm, d_counts, r->base, p2p_peers(c), c->rank, c->cap, c->status, 0);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_p2p_push_rows(mee_p2p* c, mee_router* r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts,
                      const float* d_rows, size_t n, void* stream) {
    MEE_RANGE("mee_p2p_push_rows");
    if (!c || !r || !d_counts || (n && (!d_send_keys || !d_perm))) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push_rows: null argument");
    if (!c->connected || !c->inbox_rows) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push_rows: context not connected or created without payload");
    if (r->n_shards != c->n_shards || n > c->max_batch) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_push_rows: router/batch mismatch");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    const dim3 gk(grid_for(c->cap, 256, 1024), c->n_shards);
    p2p_push_kernel<<<gk, 256, 0, st>>>(d_send_keys, d_perm, d_counts, r->base, p2p_peers(c), c->rank, c->cap, c->status, 1);
    if (d_rows) {  // null: keys only, padded (an owner-side find_or_insert over the whole inbox)
        const dim3 gr(grid_for(n / c->n_shards + 64, 16, 4096), c->n_shards);
        p2p_push_rows_kernel<<<gr, 256, 0, st>>>((const float4*)d_rows, d_perm, d_counts, r->base, p2p_peers(c), c->rank, c->cap, c->dim / 4);
    }
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_p2p_inbox(mee_p2p* c, int64_t** d_keys, float** d_rows, uint64_t* n_slots) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_inbox: null argument");
    if (d_keys) *d_keys = c->inbox_keys;
    if (d_rows) *d_rows = c->inbox_rows;
    if (n_slots) *n_slots = (uint64_t)c->n_shards * c->cap;
    return MEE_OK;
}

int mee_p2p_find(mee_p2p* c, const mee_table* t, void* stream) {
    MEE_RANGE("mee_p2p_find");
    if (!c || !t) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_find: null argument");
    if (!c->connected) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_find: not connected");
    const TableView v = table_view(t);
    if (v.dim != c->dim || v.device != c->device) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_find: table dim/device mismatch");
    DeviceGuard g(c->device);
    const dim3 grid(grid_for(c->cap, 32, 1u << 16), c->n_shards);
    hipStream_t st = (hipStream_t)stream;
#define P2PFIND(D4, RR) p2p_find_kernel<D4, RR><<<grid, 256, 0, st>>>(v.keys, (const float4*)v.values, v.nb, v.dim4, v.default_value, c->inbox_keys, \
                                                                      c->inbox_dst, c->inbox_cnt, p2p_peers(c), c->cap)
    if (v.dim4 == 16) P2PFIND(16, 2); else if (v.dim4 == 32) P2PFIND(32, 1); else P2PFIND(0, 1);
#undef P2PFIND
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_p2p_barrier(mee_p2p* c, void* stream) {
    MEE_RANGE("mee_p2p_barrier");
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_barrier: null argument");
    if (!c->connected) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_barrier: not connected");
    DeviceGuard g(c->device);
    // wall_clock64 ticks at 100 MHz: wait at most 5 s for the slowest rank
    p2p_barrier_kernel<<<1, 64, 0, (hipStream_t)stream>>>(p2p_peers(c), c->rank, c->n_shards, c->status, 500000000ull);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_p2p_status(mee_p2p* c, uint32_t* bits_out, void* stream) {
    MEE_RANGE("mee_p2p_status");
    if (!c || !bits_out) return fail(MEE_ERR_INVALID_ARG, "mee_p2p_status: null argument");
    DeviceGuard g(c->device);
    MEE_HIP(hipMemcpyAsync(bits_out, c->status, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MEE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MEE_OK;
}

int mee_scatter_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream) {
    MEE_RANGE("mee_scatter_rows");
    return permute_rows<true>(d_rows, d_perm, n, row_bytes, d_out, stream, "mee_scatter_rows");
}
int mee_gather_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream) {
    MEE_RANGE("mee_gather_rows");
    return permute_rows<false>(d_rows, d_perm, n, row_bytes, d_out, stream, "mee_gather_rows");
}

}  // extern "C"

// meepo_router.hip — hashing KATs on device, stable shard partition and row (un)permutation for gfx950.
//
// Reference anchor: /root/reference/README.md:2 ("A distributed … Embedding"); no code upstream.  Semantics:
// SPEC.md §1 (hashing) and §5 (sharding).  The partition is a three-kernel counting sort by owner that keeps
// batch order inside each shard segment: per-block histograms (wave ballot + popcount per shard), a per-shard
// scan over blocks (one wave per shard, shuffle prefix-sum), and a scatter that recomputes each key's rank
// from the same ballots.
#include <hip/hip_runtime.h>

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

__global__ __launch_bounds__(kPartBlock) void part_count_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t g,
                                                                uint32_t* blockcnt) {
    __shared__ uint32_t wcnt[kPartBlock / 64][kMaxShards];
    const uint32_t i = blockIdx.x * kPartBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool inb = i < n;
    const uint32_t o = inb ? owner_of(keys[i], g) : 0xFFFFFFFFu;
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

__global__ __launch_bounds__(kPartBlock) void part_scatter_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t g,
                                                                  const uint32_t* __restrict__ blockoff,
                                                                  const uint64_t* __restrict__ base, int64_t* send_keys,
                                                                  int64_t* perm) {
    __shared__ uint32_t wcnt[kPartBlock / 64][kMaxShards];
    const uint32_t i = blockIdx.x * kPartBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool inb = i < n;
    const int64_t key = inb ? keys[i] : 0;
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

}  // namespace mee

using namespace mee;

extern "C" {

int mee_hash_batch(const int64_t* d_keys, size_t n, uint64_t n_buckets, uint32_t n_shards, uint64_t* d_mix_out,
                   uint64_t* d_bucket_out, uint32_t* d_owner_out, void* stream) {
    if (n && !d_keys) return fail(MEE_ERR_INVALID_ARG, "mee_hash_batch: null keys");
    if (n == 0) return MEE_OK;
    hash_batch_kernel<<<grid_for(n, 256, 1u << 14), 256, 0, (hipStream_t)stream>>>(d_keys, n, n_buckets, n_shards, d_mix_out,
                                                                                 d_bucket_out, d_owner_out);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_router_destroy(mee_router* r) {
    if (!r) return MEE_OK;
    DeviceGuard g(r->device);
    if (r->blockcnt) (void)hipFree(r->blockcnt);
    if (r->base) (void)hipFree(r->base);
    delete r;
    return MEE_OK;
}

int mee_router_create(int32_t device, uint64_t max_batch, uint32_t n_shards, mee_router** out) {
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

int mee_partition(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm,
                  void* stream) {
    if (!r || !d_counts || (n && (!d_keys || !d_send_keys || !d_perm))) return fail(MEE_ERR_INVALID_ARG, "mee_partition: null argument");
    if (n > r->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_partition: n=%zu exceeds max_batch=%llu", n, (unsigned long long)r->max_batch);
    DeviceGuard g(r->device);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nblk = (uint32_t)((n + kPartBlock - 1) / kPartBlock);
    if (nblk) part_count_kernel<<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt);
    part_scan_kernel<<<1, 1024, 0, st>>>(r->blockcnt, nblk, r->n_shards, r->base, d_counts);
    if (nblk) part_scatter_kernel<<<nblk, kPartBlock, 0, st>>>(d_keys, (uint32_t)n, r->n_shards, r->blockcnt, r->base, d_send_keys, d_perm);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_scatter_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream) {
    return permute_rows<true>(d_rows, d_perm, n, row_bytes, d_out, stream, "mee_scatter_rows");
}
int mee_gather_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream) {
    return permute_rows<false>(d_rows, d_perm, n, row_bytes, d_out, stream, "mee_gather_rows");
}

}  // extern "C"

// meepo_sharded.hip — the row-sharded exchange behind the C-ABI (SPEC.md §5): one context per rank, RCCL over xGMI.
//
// Reference anchor: /root/reference/README.md:2 ("A distributed … Embedding"); the snapshot has no code.  SURVEY.md §8b
// ("sharded variants take a communicator handle"), §8e ("RCCL all-to-all(v) ×2 per direction").
//
// Per sharded operator, all on the caller's stream:
//     mee_partition (stable counting sort by owner)                                  -> send order, counts[G], perm
//     keys (+ value / gradient rows) to their owners:  ncclGroupStart; G x ncclSend / ncclRecv; ncclGroupEnd
//     the local table's operator on what arrived (ordered by source rank, then batch position: last-wins across ranks)
//     rows + found bytes back to the requesters, ONE grouped exchange                -> un-permute kernel into batch order
//
// Two segment layouts:
//   exact  (pad_slack = 0): a counts exchange (8 B per peer) and ONE host synchronisation give every message its exact size.
//   padded (pad_slack > 0): every (source, owner) segment has the fixed capacity cap = max_batch / G * pad_slack + 1024 and is
//          padded with EMPTY keys (= padding, skipped by every operator, SPEC.md §2): message sizes are constants, nothing
//          returns to the host, the launch sequence is static.  A segment that would exceed cap drops keys and sets bit 0 of
//          mee_sharded_status (uniform hashing of 1M keys over 8 owners deviates by < 1 %; skewed batches: de-duplicate first,
//          or use the exact layout).
//
// Options of a context (mee_sharded_create_ex): a COLD table behind the local one (BASELINE configs[4]: every shard a hot/cold pair — the
// owner side runs mee_find + mee_find_missing and sends mutators to both tiers, as meepoembedding_amd/tiered.py does), and pre-exchange
// DEDUP (SURVEY.md §7 lever (a)): lookups send only a batch's distinct keys (every occurrence is then served from its key's row), optimizer
// applies send ONE (key, summed gradient row) pair per distinct key of the rank's batch (mee_dedup_sum: fp64 sums rounded once) — on a
// skewed stream the bytes on xGMI scale with the distinct keys, in both directions of a training step.
//
// Errors inside a collective operator: owner-side buffers are sized once, at creation, for the most that can arrive (G x max_batch, the same
// max_batch on every rank — checked collectively), so no rank can fail for memory between two exchange steps; an RCCL error marks the
// context dead and aborts the communicator (ncclCommAbort) so that peers do not wait for this rank forever.
//
// RCCL is bound at first use with dlopen("librccl.so.1"): a single-GPU process never loads it, and under PyTorch the
// soname resolves to the copy torch has already loaded (one RCCL per process).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <mutex>
#include <unordered_set>
#include <vector>

#include "meepo_device.h"
#include "meepo_host.h"

namespace mee {

// ---- RCCL entry points, bound lazily ------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: older builds / the test stand-in may lack it
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char error[256] = {0};
};

static RcclApi& rccl_state() {
    static RcclApi api;
    return api;
}
static RcclApi* rccl_api() {
    RcclApi& api = rccl_state();
    static std::once_flag once;
    std::call_once(once, [&api] {
        // MEE_RCCL_LIB names the library to bind instead (a particular RCCL build; the multi-rank-on-one-GPU transport of the test
        // suite, tests/cabi/fake_rccl.cpp): when set it is the only candidate, so a typo fails loudly instead of binding the default
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char* forced = getenv("MEE_RCCL_LIB")) {
            api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) { snprintf(api.error, sizeof api.error, "dlopen(MEE_RCCL_LIB=%s): %s", forced, dlerror()); return; }
        } else
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { snprintf(api.error, sizeof api.error, "dlopen(librccl.so.1): %s", dlerror()); return; }
        bool ok = true;
#define BIND(field, sym) do { *(void**)(&api.field) = dlsym(api.handle, sym); if (!api.field) { ok = false; snprintf(api.error, sizeof api.error, "librccl: symbol %s missing", sym); } } while (0)
        BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
        BIND(CommCount, "ncclCommCount"); BIND(CommUserRank, "ncclCommUserRank"); BIND(CommCuDevice, "ncclCommCuDevice");
        BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv");
        BIND(AllReduce, "ncclAllReduce"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
        *(void**)(&api.CommAbort) = dlsym(api.handle, "ncclCommAbort");
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return api.handle ? &api : nullptr;
}
static const char* rccl_load_error() {
    (void)rccl_api();
    return rccl_state().error[0] ? rccl_state().error : "RCCL could not be loaded (librccl.so.1 not found or incomplete)";
}

#define MEE_NCCL(api, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess)                                                                                     \
            return ::mee::fail(MEE_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---- kernels: segment packing / un-permute -------------------------------------------------------------------------
// A "segment" is what one requester sends to one owner.  Exact layout: segments are contiguous in partition order,
// segment p starts at base[p] = counts[0] + … + counts[p-1].  Padded layout: segment p starts at p * cap and holds cap positions.
__device__ __forceinline__ uint64_t seg_base(const uint64_t* __restrict__ counts, uint32_t p) {
    uint64_t b = 0;
    for (uint32_t q = 0; q < p; ++q) b += counts[q];
    return b;
}

// padded layout, keys: pad[p*cap + j] = j < counts[p] ? send_keys[base[p] + j] : EMPTY      grid (x, G)
__global__ __launch_bounds__(256) void shard_pad_keys_kernel(const int64_t* __restrict__ send_keys, const uint64_t* __restrict__ counts,
                                                             uint64_t cap, int64_t* __restrict__ pad, uint32_t* status) {
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p], take = cnt < cap ? cnt : cap, b0 = seg_base(counts, p);
    int64_t* __restrict__ dst = pad + (uint64_t)p * cap;
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < cap; j += (uint64_t)gridDim.x * blockDim.x)
        dst[j] = j < take ? send_keys[b0 + j] : kEmpty;
    if (cnt > cap && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1u);
}

// payload rows into send order: dst[off(p) + j] = rows[perm[base[p] + j]]; one 16-lane tile per row      grid (x, G)
__global__ __launch_bounds__(256) void shard_pack_rows_kernel(const float4* __restrict__ rows, const int64_t* __restrict__ perm,
                                                              const uint64_t* __restrict__ counts, uint64_t cap /* 0 = exact */,
                                                              uint32_t dim4, float4* __restrict__ dst) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p], take = (cap && cnt > cap) ? cap : cnt, b0 = seg_base(counts, p);
    const uint64_t off = cap ? (uint64_t)p * cap : b0;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t j0 = wave * 4; j0 < take; j0 += n_waves * 4) {
        const uint64_t j = j0 + tile;
        if (j >= take) continue;
        const uint64_t src = (uint64_t)perm[b0 + j] * dim4;
        for (uint32_t c = tl; c < dim4; c += 16) dst[(off + j) * dim4 + c] = rows[src + c];
    }
}

// what came back, into batch order: out[perm[base[p] + j]] = back_rows[off(p) + j], found likewise (either nullable)   grid (x, G)
// Padded layout: positions j >= cap of a segment were never sent (the segment overflowed: status bit 0): they get the default row and
// found = 0 — a dropped lookup must not hand the caller uninitialised memory.  The owner's find skips MEE_EMPTY_KEY positions as padding
// (no row, no found byte), so a reserved key the CALLER put into its batch is answered here, on the requester side, with the default row.
__global__ __launch_bounds__(256) void shard_return_kernel(const float4* __restrict__ back_rows, const uint8_t* __restrict__ back_found,
                                                           const int64_t* __restrict__ perm, const uint64_t* __restrict__ counts,
                                                           uint64_t cap /* 0 = exact */, uint32_t dim4, float4* __restrict__ out,
                                                           uint8_t* __restrict__ found, float defv, const int64_t* __restrict__ send_keys) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p], take = (cap && cnt > cap) ? cap : cnt, b0 = seg_base(counts, p);
    const uint64_t off = cap ? (uint64_t)p * cap : b0;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const float4 def4 = make_float4(defv, defv, defv, defv);
    for (uint64_t j0 = wave * 4; j0 < cnt; j0 += n_waves * 4) {
        const uint64_t j = j0 + tile;
        if (j >= cnt) continue;
        const uint64_t dstp = (uint64_t)perm[b0 + j];
        const bool sent = j < take && !(cap && reserved_key(send_keys[b0 + j]));
        if (out && (back_rows || !sent))
            for (uint32_t c = tl; c < dim4; c += 16) out[dstp * dim4 + c] = sent ? back_rows[(off + j) * dim4 + c] : def4;
        if (found && (back_found || !sent) && tl == 0) found[dstp] = sent ? back_found[off + j] : (uint8_t)0;
    }
}

// pre-exchange dedup, last step: every occurrence takes the row of its distinct key: out[i] = urows[inverse[i]], found likewise
// (four positions in flight per 16-lane tile: index, then row, then a streamed store — with one position per tile the kernel was a chain of two dependent loads per
// row at 3.3 TB/s: 108 us per 1M dim-64 rows)
__global__ __launch_bounds__(256) void shard_expand_kernel(const float4* __restrict__ urows, const uint8_t* __restrict__ ufound,
                                                           const int64_t* __restrict__ inverse, uint64_t n, uint32_t dim4,
                                                           float4* __restrict__ out, uint8_t* __restrict__ found) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t i0 = wave * 16; i0 < n; i0 += n_waves * 16) {
        uint64_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint64_t i = i0 + q * 4 + tile; u[q] = i < n ? (uint64_t)inverse[i] : 0ull; }
        for (uint32_t c = tl; c < dim4; c += 16) {
            float4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = urows[u[q] * dim4 + c];   // (cached loads: a skewed batch reads its hot rows many times)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t i = i0 + q * 4 + tile;
                if (i < n) __builtin_nontemporal_store(f32x4{r[q].x, r[q].y, r[q].z, r[q].w}, reinterpret_cast<f32x4*>(out) + i * dim4 + c);
            }
        }
        if (found && tl < 4) {   // lane tl of the tile: the found byte of the tile's position q = tl
            const uint64_t i = i0 + (uint64_t)tl * 4 + tile;
            if (i < n) found[i] = ufound[tl == 0 ? u[0] : tl == 1 ? u[1] : tl == 2 ? u[2] : u[3]];
        }
    }
}

__global__ void shard_or_kernel(uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n) {   // a |= b (found masks of the two tiers)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] |= b[i];
}
__global__ void shard_fill_row_kernel(float* __restrict__ row, uint32_t dim, float v, uint8_t* found_byte) {
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) row[i] = v;
    if (threadIdx.x == 0) *found_byte = 0;
}
// chunked apply: a key -> mix64(key), padding stays padding.  Partitioning the images by owner_of() splits a shard's arrivals by bits that
// are independent of the bits that sent them to this shard.
__global__ void shard_rehash_kernel(const int64_t* __restrict__ keys, int64_t* __restrict__ out, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t k = keys[i];
        int64_t h = (int64_t)mix64((uint64_t)k);
        if (reserved_key(h)) h = 0;   // an image that lands on a reserved value: any ordinary value will do (it only picks the chunk)
        out[i] = reserved_key(k) ? kEmpty : h;
    }
}

}  // namespace mee

using namespace mee;

struct mee_sharded {
    int device;
    uint32_t G, rank, dim, dim4;
    uint64_t max_batch;       // lookups / pairs per call of THIS rank (the same on every rank: checked at creation)
    uint64_t cap;             // padded layout: positions per (source, owner) segment; 0 = exact layout
    uint64_t local_max_batch; // the local table's config.max_batch (bounds one local mutator call)
    uint32_t optimizer;
    float defv;               // the table's default row value (rows of lookups a padded segment dropped)
    bool dead;                // an RCCL call failed inside an operator: the communicator was aborted, every later call fails at once
    mee_table* local;         // the shard (the HOT table of a tiered shard)
    mee_table* cold;          // nullable: the cold tier behind it (rows in pinned host DRAM)
    uint64_t hot_limit, hot_ub;   // tiered: new keys go hot while the hot table holds fewer keys than hot_limit; hot_ub = upper bound on its size
    mee_router* router;
    ncclComm_t comm;
    // requester side
    int64_t *send_keys, *perm, *pad_keys;
    uint64_t* counts;          // [G] device: keys this rank sends to each owner
    uint64_t* d_recv_counts;   // [max(G, 64)] device: keys each source sends here (exact layout); scratch of the collective checks
    uint64_t* h_counts;        // [2 * max(G, 64)] pinned: counts, recv counts
    float* send_rows;          // payload rows in send order (allocated by the first mutator call)
    float* back_rows;          // rows returned by the owners, in send order
    uint8_t* back_found;
    uint64_t send_slots;       // positions the requester-side buffers hold: max_batch (exact) or G*cap (padded)
    // pre-exchange dedup (MEE_SHARDED_DEDUP): a scratch-only table of our own groups the batch's keys (so that contexts sharing one
    // shard may keep lookups in flight side by side); distinct keys, each position's index into them, their rows in "unique order"
    // (lookups: the rows that came back; applies: the rank's summed gradient rows before they leave)
    bool dedup;
    mee_table* dd;
    int64_t *uniq, *inverse;
    float* urows;              // [max_batch + 1][dim]: row max_batch = the default row (reserved keys point at it)
    uint8_t* ufound;           // [max_batch + 1]
    // owner side: sized once for the most that can arrive, G x max_batch (exact) or G x cap (padded)
    uint64_t recv_slots;
    int64_t* recv_keys;
    float* recv_rows;          // payload received / rows found
    uint8_t *recv_found, *recv_found2;   // second mask: the cold tier's answer (OR-ed into the first)
    uint32_t* status;          // device: bit 0 = a padded segment overflowed
    // chunked apply (exact layout, arrivals beyond the local table's max_batch): made on first need
    mee_router* sub_router; uint32_t sub_chunks;
    int64_t *img_keys, *keys2, *perm2; uint64_t* counts2; float* rows2; uint64_t* h_counts2;
    std::vector<size_t> scount, sdisp, rcount, rdisp;   // in positions
};

namespace mee {

static void sharded_free(mee_sharded* c) {
    void* dev[] = {c->send_keys, c->perm, c->pad_keys, c->counts, c->d_recv_counts, c->send_rows, c->back_rows, c->back_found,
                   c->recv_keys, c->recv_rows, c->recv_found, c->recv_found2, c->status, c->uniq, c->inverse, c->urows, c->ufound,
                   c->img_keys, c->keys2, c->perm2, c->counts2, c->rows2};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_counts2) (void)hipHostFree(c->h_counts2);
    if (c->router) mee_router_destroy(c->router);
    if (c->sub_router) mee_router_destroy(c->sub_router);
    if (c->dd) mee_table_destroy(c->dd);
}

static int ensure_send_rows(mee_sharded* c) {
    if (c->send_rows) return MEE_OK;
    MEE_HIP(hipMalloc((void**)&c->send_rows, c->send_slots * (uint64_t)c->dim * 4));
    return MEE_OK;
}

struct Leg {              // one buffer pair of a grouped exchange
    const void* send; void* recv;
    size_t elems;         // elements per position
    size_t elem_bytes;
    ncclDataType_t dt;
};

// Communicators this library has aborted.  A context only BORROWS its communicator (the caller's own ncclComm_t, or one made by
// mee_comm_create and shared by several contexts), and ncclCommAbort FREES it: whoever owns it must not destroy it afterwards, and no other
// context on it may issue another call.  So the abort is recorded per communicator, process-wide: every context on an aborted communicator
// fails at once (check_call), mee_comm_destroy of an aborted communicator does nothing, and an owner that made the communicator itself asks
// mee_comm_aborted() before it calls ncclCommDestroy.
static std::mutex g_abort_mu;
static std::unordered_set<void*> g_aborted;
static bool comm_is_aborted(void* comm) { std::lock_guard<std::mutex> lk(g_abort_mu); return g_aborted.count(comm) != 0; }
static void abort_comm(RcclApi* api, ncclComm_t comm) {
    {
        std::lock_guard<std::mutex> lk(g_abort_mu);
        if (!g_aborted.insert((void*)comm).second) return;   // another context on it was first: ncclCommAbort runs once
    }
    if (api->CommAbort) (void)api->CommAbort(comm);
}

// an RCCL call failed inside a collective operator: peers may be waiting for this rank inside the same group.  Abort the communicator
// (their calls then return with an error instead of hanging) and refuse every later call on this context and on every other context that
// shares the communicator.
static int rccl_failed(mee_sharded* c, RcclApi* api, ncclResult_t err, const char* what) {
    c->dead = true;
    abort_comm(api, c->comm);
    return fail(MEE_ERR_RCCL, "%s failed: %s — the communicator was aborted (and thereby freed: do not destroy it), every sharded context on it is unusable", what, api->GetErrorString(err));
}

// forward: requester segments (scount/sdisp) -> owner segments (rcount/rdisp); reverse: the other way round.
// ONE ncclGroup for all legs and peers; the segment a rank keeps for itself is a device-to-device copy.
static int exchange(mee_sharded* c, const Leg* legs, int n_legs, bool reverse, hipStream_t st) {
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    const std::vector<size_t>& sc = reverse ? c->rcount : c->scount;
    const std::vector<size_t>& sd = reverse ? c->rdisp : c->sdisp;
    const std::vector<size_t>& rc = reverse ? c->scount : c->rcount;
    const std::vector<size_t>& rd = reverse ? c->sdisp : c->rdisp;
    for (int l = 0; l < n_legs; ++l) {
        const size_t row = legs[l].elems * legs[l].elem_bytes;
        if (sc[c->rank])
            MEE_HIP(hipMemcpyAsync((char*)legs[l].recv + rd[c->rank] * row, (const char*)legs[l].send + sd[c->rank] * row, sc[c->rank] * row,
                                   hipMemcpyDeviceToDevice, st));
    }
    if (c->G == 1) return MEE_OK;
    ncclResult_t first_err = api->GroupStart();
    if (first_err != ncclSuccess) return rccl_failed(c, api, first_err, "ncclGroupStart");
    for (uint32_t p = 0; p < c->G && first_err == ncclSuccess; ++p) {   // an open group must be closed whatever happens inside it
        if (p == c->rank) continue;
        for (int l = 0; l < n_legs && first_err == ncclSuccess; ++l) {
            const size_t row = legs[l].elems * legs[l].elem_bytes;
            // zero-length messages are skipped on both sides (the two ends agree on every count)
            if (sc[p]) first_err = api->Send((const char*)legs[l].send + sd[p] * row, sc[p] * legs[l].elems, legs[l].dt, (int)p, c->comm, st);
            if (rc[p] && first_err == ncclSuccess) first_err = api->Recv((char*)legs[l].recv + rd[p] * row, rc[p] * legs[l].elems, legs[l].dt, (int)p, c->comm, st);
        }
    }
    const ncclResult_t end_err = api->GroupEnd();
    if (first_err != ncclSuccess) return rccl_failed(c, api, first_err, "ncclSend/ncclRecv");
    if (end_err != ncclSuccess) return rccl_failed(c, api, end_err, "ncclGroupEnd");
    return MEE_OK;
}

// partition + (exact) counts exchange and the one host synchronisation | (padded) pad the key segments.
// On return keys_to_send points at the keys in segment layout and *r_total = positions the owner side will hold.
// skip_padding: d_keys may hold MEE_EMPTY_KEY padding that belongs to no shard (the padded unique list of a de-duplicated batch).
static int route(mee_sharded* c, const int64_t* d_keys, size_t n, hipStream_t st, const int64_t** keys_to_send, uint64_t* r_total, bool skip_padding = false) {
    if (int rc = skip_padding ? mee_partition_padded(c->router, d_keys, n, c->send_keys, c->counts, c->perm, st)
                              : mee_partition(c->router, d_keys, n, c->send_keys, c->counts, c->perm, st)) return rc;
    if (c->cap) {
        const dim3 grid(grid_for(c->cap, 256, 256), c->G);
        shard_pad_keys_kernel<<<grid, 256, 0, st>>>(c->send_keys, c->counts, c->cap, c->pad_keys, c->status);
        MEE_HIP(hipGetLastError());
        *keys_to_send = c->pad_keys;
        *r_total = (uint64_t)c->G * c->cap;
        return MEE_OK;
    }
    // exact layout: every owner learns how many keys each source sends (8 B per peer, one grouped exchange on the device) …
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    MEE_HIP(hipMemcpyAsync(c->d_recv_counts + c->rank, c->counts + c->rank, 8, hipMemcpyDeviceToDevice, st));
    if (c->G > 1) {
        ncclResult_t first_err = api->GroupStart();
        if (first_err != ncclSuccess) return rccl_failed(c, api, first_err, "ncclGroupStart");
        for (uint32_t p = 0; p < c->G && first_err == ncclSuccess; ++p) {
            if (p == c->rank) continue;
            first_err = api->Send(c->counts + p, 1, ncclUint64, (int)p, c->comm, st);
            if (first_err == ncclSuccess) first_err = api->Recv(c->d_recv_counts + p, 1, ncclUint64, (int)p, c->comm, st);
        }
        const ncclResult_t end_err = api->GroupEnd();
        if (first_err != ncclSuccess || end_err != ncclSuccess) return rccl_failed(c, api, first_err != ncclSuccess ? first_err : end_err, "counts exchange");
    }
    // … and both count vectors reach the host: the ONE synchronisation of an exact-layout operator
    MEE_HIP(hipMemcpyAsync(c->h_counts, c->counts, c->G * 8, hipMemcpyDeviceToHost, st));
    MEE_HIP(hipMemcpyAsync(c->h_counts + c->G, c->d_recv_counts, c->G * 8, hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    size_t s_acc = 0, r_acc = 0;
    for (uint32_t p = 0; p < c->G; ++p) {
        c->scount[p] = (size_t)c->h_counts[p]; c->sdisp[p] = s_acc; s_acc += c->scount[p];
        c->rcount[p] = (size_t)c->h_counts[c->G + p]; c->rdisp[p] = r_acc; r_acc += c->rcount[p];
    }
    if (r_acc > c->recv_slots) {   // cannot happen when every rank keeps to the max_batch all ranks agreed on: a peer sent more than it may
        c->dead = true;
        abort_comm(api, c->comm);
        return fail(MEE_ERR_BATCH_TOO_LARGE, "%llu keys arrive at this shard, more than G x max_batch = %llu: a rank exceeded the max_batch the ranks agreed on",
                    (unsigned long long)r_acc, (unsigned long long)c->recv_slots);
    }
    *keys_to_send = c->send_keys;
    *r_total = r_acc;
    return MEE_OK;
}

static int pack_rows(mee_sharded* c, const float* d_rows, hipStream_t st) {
    if (int rc = ensure_send_rows(c)) return rc;
    const dim3 grid(grid_for(c->max_batch / c->G + 64, 16, 4096), c->G);
    shard_pack_rows_kernel<<<grid, 256, 0, st>>>((const float4*)d_rows, c->perm, c->counts, c->cap, c->dim4, (float4*)c->send_rows);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// rows / found bytes back to the requesters, then into batch order (`d_out` / `d_found`: caller buffers, or the unique-order buffers of a
// de-duplicated lookup)
static int give_back(mee_sharded* c, bool rows, float* d_out, uint8_t* d_found, hipStream_t st) {
    Leg legs[2];
    int nl = 0;
    if (rows) legs[nl++] = Leg{c->recv_rows, c->back_rows, c->dim, 4, ncclFloat};
    legs[nl++] = Leg{c->recv_found, c->back_found, 1, 1, ncclUint8};
    if (int rc = exchange(c, legs, nl, /*reverse=*/true, st)) return rc;
    const dim3 grid(grid_for(c->max_batch / c->G + 64, 16, 4096), c->G);
    shard_return_kernel<<<grid, 256, 0, st>>>(rows ? (const float4*)c->back_rows : nullptr, c->back_found, c->perm, c->counts, c->cap, c->dim4,
                                              (float4*)d_out, d_found, c->defv, c->send_keys);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

static int check_call(const mee_sharded* c, size_t n, const char* name) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "%s: null context", name);
    if (c->dead || comm_is_aborted((void*)c->comm)) return fail(MEE_ERR_RCCL, "%s: an earlier RCCL error aborted this context's communicator", name);
    if (n > c->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: n=%zu exceeds the context's max_batch=%llu", name, n, (unsigned long long)c->max_batch);
    return MEE_OK;
}

// keys (+ payload rows) to their owners; on return recv_keys / recv_rows [0, *r_total) hold what arrived
static int push(mee_sharded* c, const int64_t* d_keys, const float* d_rows, size_t n, hipStream_t st, uint64_t* r_total, bool skip_padding = false) {
    const int64_t* ks = nullptr;
    if (int rc = route(c, d_keys, n, st, &ks, r_total, skip_padding)) return rc;
    Leg legs[2];
    int nl = 0;
    legs[nl++] = Leg{ks, c->recv_keys, 1, 8, ncclInt64};
    if (d_rows) {
        if (int rc = pack_rows(c, d_rows, st)) return rc;
        legs[nl++] = Leg{c->send_rows, c->recv_rows, c->dim, 4, ncclFloat};
    }
    return exchange(c, legs, nl, /*reverse=*/false, st);
}

// ---- owner side: the local operator over what arrived; a tiered shard = the hot table in front of the cold one (tiered.py in C++) ----------
// does the hot table have room for up to n new keys?  (an upper bound kept on the host; refreshed — one mee_size, which synchronises — only
// when the bound is hit)
static int hot_has_room(mee_sharded* c, uint64_t n, void* stream, bool* room) {
    if (c->hot_ub + n > c->hot_limit) {
        size_t sz = 0;
        if (int rc = mee_size(c->local, &sz, stream)) return rc;
        c->hot_ub = sz;
    }
    *room = c->hot_ub + n <= c->hot_limit;
    return MEE_OK;
}

static int owner_lookup(mee_sharded* c, uint64_t rt, bool insert_missing, void* stream) {
    if (rt == 0) return MEE_OK;
    if (!c->cold && insert_missing) {
        for (uint64_t s = 0; s < rt; s += c->local_max_batch) {   // find_or_insert is sequentially consistent: chunks are fine
            const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
            if (int rc = mee_find_or_insert(c->local, c->recv_keys + s, m, c->recv_rows + s * c->dim, c->recv_found + s, stream)) return rc;
        }
        return MEE_OK;
    }
    // padded segments: EMPTY positions are padding nobody reads — the find writes neither a default row nor a found byte for them
    if (int rc = c->cap ? find_skip_padding(c->local, c->recv_keys, rt, c->recv_rows, c->recv_found, stream)
                        : mee_find(c->local, c->recv_keys, rt, c->recv_rows, c->recv_found, stream)) return rc;
    if (!c->cold) return MEE_OK;
    if (int rc = mee_find_missing(c->cold, c->recv_keys, rt, c->recv_rows, c->recv_found, stream)) return rc;   // second tier, same buffers, no sync
    if (!insert_missing) return MEE_OK;
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {   // keys in neither tier are created: hot while there is room, else cold
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        bool room = false;
        if (int rc = hot_has_room(c, m, stream, &room)) return rc;
        if (int rc = mee_find_or_insert_missing(room ? c->local : c->cold, c->recv_keys + s, m, c->recv_rows + s * c->dim, c->recv_found + s, stream)) return rc;
        if (room) c->hot_ub += m;
    }
    return MEE_OK;
}

static int owner_insert(mee_sharded* c, uint64_t rt, void* stream) {
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {   // upserts are sequentially consistent: order kept = last-wins kept
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        const int64_t* k = c->recv_keys + s;
        const float* v = c->recv_rows + s * c->dim;
        if (!c->cold) { if (int rc = mee_insert(c->local, k, v, m, stream)) return rc; continue; }
        // a key lives in exactly one tier: overwrite it where it is (both tiers see the chunk, each ignores keys it does not hold), create
        // the keys neither holds in the hot tier while it has room, else in the cold one
        if (int rc = mee_assign(c->local, k, v, m, c->recv_found + s, stream)) return rc;
        if (int rc = mee_assign(c->cold, k, v, m, c->recv_found2 + s, stream)) return rc;
        shard_or_kernel<<<grid_for(m, 256, 2048), 256, 0, (hipStream_t)stream>>>(c->recv_found + s, c->recv_found2 + s, m);
        bool room = false;
        if (int rc = hot_has_room(c, m, stream, &room)) return rc;
        if (int rc = mee_insert_missing(room ? c->local : c->cold, k, v, m, c->recv_found + s, stream)) return rc;
        if (room) c->hot_ub += m;
    }
    return MEE_OK;
}

static int owner_assign_or_remove(mee_sharded* c, uint64_t rt, bool assign, void* stream) {
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        const int64_t* k = c->recv_keys + s;
        const float* v = c->recv_rows + s * c->dim;
        if (int rc = assign ? mee_assign(c->local, k, v, m, c->recv_found + s, stream) : mee_remove(c->local, k, m, c->recv_found + s, stream)) return rc;
        if (c->cold) {
            if (int rc = assign ? mee_assign(c->cold, k, v, m, c->recv_found2 + s, stream) : mee_remove(c->cold, k, m, c->recv_found2 + s, stream)) return rc;
            shard_or_kernel<<<grid_for(m, 256, 2048), 256, 0, (hipStream_t)stream>>>(c->recv_found + s, c->recv_found2 + s, m);
        }
    }
    return MEE_OK;
}

struct ApplySpec { bool adam; float lr, beta1, beta2, eps; uint64_t step; };
static int apply_tiers(mee_sharded* c, const int64_t* k, const float* g, uint64_t n, const ApplySpec& a, void* stream) {
    mee_table* tiers[2] = {c->local, c->cold};
    for (mee_table* t : tiers) {   // a key lives in one tier and each table ignores keys it does not hold: both see the whole batch
        if (!t) continue;
        if (int rc = a.adam ? mee_apply_adam(t, k, g, n, a.lr, a.beta1, a.beta2, a.eps, a.step, stream) : mee_apply_adagrad(t, k, g, n, a.lr, a.eps, stream)) return rc;
    }
    return MEE_OK;
}

// More pairs arrived than one local apply takes (exact layout: a skewed step, or simply G ranks x max_batch against a table made for one
// rank's batch).  An apply is ONE update per distinct key, so the arrivals are split BY KEY: every key's pairs go to one chunk (chunk =
// owner_of(mix64(key), n_chunks): bits independent of the ones that brought the keys to this shard), each chunk is an apply of its own.
// One more host synchronisation (the chunk sizes); buffers are made on first need.
static int chunked_apply(mee_sharded* c, uint64_t rt, const ApplySpec& a, void* stream, const char* name) {
    hipStream_t st = (hipStream_t)stream;
    const uint64_t per = c->local_max_batch * 3 / 4 ? c->local_max_batch * 3 / 4 : 1;   // aim at chunks of 3/4 of what one apply takes: room for uneven chunks
    uint32_t want = (uint32_t)((rt + per - 1) / per);
    want = want < 2 ? 2 : want > 64 ? 64 : want;
    if (!c->sub_router || c->sub_chunks < want) {
        MEE_HIP(hipStreamSynchronize(st));
        if (c->sub_router) { mee_router_destroy(c->sub_router); c->sub_router = nullptr; }
        if (int rc = mee_router_create(c->device, c->recv_slots, want, &c->sub_router)) return rc;
        c->sub_chunks = want;
        if (!c->keys2) {
            MEE_HIP(hipMalloc((void**)&c->img_keys, c->recv_slots * 8)); MEE_HIP(hipMalloc((void**)&c->keys2, c->recv_slots * 8));
            MEE_HIP(hipMalloc((void**)&c->perm2, c->recv_slots * 8)); MEE_HIP(hipMalloc((void**)&c->counts2, 64 * 8));
            MEE_HIP(hipMalloc((void**)&c->rows2, c->recv_slots * (uint64_t)c->dim * 4));
            MEE_HIP(hipHostMalloc((void**)&c->h_counts2, 64 * 8));
        }
    }
    shard_rehash_kernel<<<grid_for(rt, 256, 4096), 256, 0, st>>>(c->recv_keys, c->img_keys, rt);
    MEE_HIP(hipGetLastError());
    if (int rc = mee_partition_padded(c->sub_router, c->img_keys, rt, c->keys2 /* images in chunk order: not used */, c->counts2, c->perm2, stream)) return rc;
    MEE_HIP(hipMemcpyAsync(c->h_counts2, c->counts2, c->sub_chunks * 8, hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    uint64_t valid = 0;
    for (uint32_t j = 0; j < c->sub_chunks; ++j) {
        if (c->h_counts2[j] > c->local_max_batch)
            return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: %llu pairs arrived at this shard and one key-range chunk still holds %llu, more than the local table's max_batch=%llu",
                        name, (unsigned long long)rt, (unsigned long long)c->h_counts2[j], (unsigned long long)c->local_max_batch);
        valid += c->h_counts2[j];
    }
    if (int rc = mee_gather_rows(c->recv_keys, c->perm2, valid, 8, c->keys2, stream)) return rc;             // the keys themselves, in chunk order
    if (int rc = mee_gather_rows(c->recv_rows, c->perm2, valid, (size_t)c->dim * 4, c->rows2, stream)) return rc;
    uint64_t off = 0;
    for (uint32_t j = 0; j < c->sub_chunks; ++j) {
        const uint64_t m = c->h_counts2[j];
        if (m) if (int rc = apply_tiers(c, c->keys2 + off, c->rows2 + off * c->dim, m, a, stream)) return rc;
        off += m;
    }
    return MEE_OK;
}

}  // namespace mee

extern "C" {

int mee_comm_unique_id(void* id_out) {
    MEE_RANGE("mee_comm_unique_id");
    if (!id_out) return fail(MEE_ERR_INVALID_ARG, "mee_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == MEE_COMM_ID_BYTES, "ncclUniqueId size");
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    ncclUniqueId id;
    MEE_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return MEE_OK;
}

int mee_comm_create(const void* id, uint32_t n_ranks, uint32_t rank, int32_t device, void** comm_out) {
    MEE_RANGE("mee_comm_create");
    if (!id || !comm_out || n_ranks == 0 || rank >= n_ranks) return fail(MEE_ERR_INVALID_ARG, "mee_comm_create: bad argument");
    *comm_out = nullptr;
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    DeviceGuard g(device);
    if (g.err != hipSuccess) return fail(MEE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(g.err));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    MEE_NCCL(api, api->CommInitRank(&comm, (int)n_ranks, uid, (int)rank));
    { std::lock_guard<std::mutex> lk(g_abort_mu); g_aborted.erase((void*)comm); }   // (an address that an aborted communicator once had)
    *comm_out = comm;
    return MEE_OK;
}

int mee_comm_destroy(void* comm) {
    MEE_RANGE("mee_comm_destroy");
    if (!comm) return MEE_OK;
    {   // ncclCommAbort has freed an aborted communicator already: forget it, there is nothing left to destroy
        std::lock_guard<std::mutex> lk(g_abort_mu);
        if (g_aborted.erase(comm)) return MEE_OK;
    }
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    MEE_NCCL(api, api->CommDestroy((ncclComm_t)comm));
    return MEE_OK;
}

int mee_comm_aborted(void* comm) { return comm && comm_is_aborted(comm) ? 1 : 0; }

int mee_sharded_destroy(mee_sharded* c) {
    MEE_RANGE("mee_sharded_destroy");
    if (!c) return MEE_OK;
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    sharded_free(c);
    delete c;
    return MEE_OK;
}

int mee_sharded_create_ex(mee_table* local, void* nccl_comm, const mee_sharded_options* opt, mee_sharded** out) {
    MEE_RANGE("mee_sharded_create_ex");
    if (!out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: null out");
    *out = nullptr;
    if (!opt || opt->struct_size != sizeof(mee_sharded_options))
        return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: null options or struct_size != %zu (ABI mismatch)", sizeof(mee_sharded_options));
    const uint64_t max_batch = opt->max_batch;
    const double pad_slack = opt->pad_slack;
    if (!local || !nccl_comm || max_batch == 0 || max_batch > (1ull << 30) || pad_slack < 0.0 || (pad_slack > 0.0 && pad_slack < 1.0) ||
        (opt->flags & ~(uint32_t)MEE_SHARDED_DEDUP))
        return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: null table / communicator, max_batch not in [1, 2^30], pad_slack not 0 or >= 1, or unknown flags");
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    int nranks = 0, rank = 0, cdev = -1;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    // A caller-owned communicator that this library once aborted is FREED; the allocator may hand its address to the caller's next ncclCommInitRank.  Whoever creates
    // a context on an address vouches that a live communicator is there now: the aborted mark of the address's previous tenant goes (it would otherwise fail every
    // context on the new communicator with MEE_ERR_RCCL for ever, and the registry would only ever grow).
    { std::lock_guard<std::mutex> lk(g_abort_mu); g_aborted.erase((void*)comm); }
    MEE_NCCL(api, api->CommCount(comm, &nranks));
    MEE_NCCL(api, api->CommUserRank(comm, &rank));
    MEE_NCCL(api, api->CommCuDevice(comm, &cdev));
    const TableView v = table_view(local);
    if (cdev != v.device) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: communicator lives on device %d, the table on device %d", cdev, v.device);
    if (nranks < 1 || nranks > 64) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: %d ranks (1..64 supported)", nranks);
    mee_table_info info, cinfo;
    if (int rc = mee_table_info_get(local, &info)) return rc;
    if (opt->cold) {
        const TableView cv = table_view(opt->cold);
        if (int rc = mee_table_info_get(opt->cold, &cinfo)) return rc;
        if (cv.device != v.device || cinfo.dim != info.dim || cinfo.optimizer != info.optimizer || opt->cold == local)
            return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: the cold table must be another table on the same device with the same dim and optimizer");
    }
    DeviceGuard g(v.device);
    mee_sharded* c = new (std::nothrow) mee_sharded();
    if (!c) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = v.device; c->G = (uint32_t)nranks; c->rank = (uint32_t)rank; c->dim = v.dim; c->dim4 = v.dim4; c->defv = v.default_value;
    c->max_batch = max_batch; c->local = local; c->comm = comm; c->optimizer = info.optimizer;
    c->cold = opt->cold;
    c->local_max_batch = opt->cold && cinfo.max_batch < info.max_batch ? cinfo.max_batch : info.max_batch;
    c->hot_limit = opt->hot_key_limit ? opt->hot_key_limit : info.capacity / 4 * 3;
    c->hot_ub = c->hot_limit + 1;   // unknown: the first creation of keys asks the table
    c->dedup = (opt->flags & MEE_SHARDED_DEDUP) != 0;
    c->cap = pad_slack > 0.0 ? (uint64_t)std::ceil((double)max_batch / nranks * pad_slack) + 1024 : 0;
    c->send_slots = c->cap ? (uint64_t)c->G * c->cap : max_batch;
    c->recv_slots = c->cap ? (uint64_t)c->G * c->cap : (uint64_t)c->G * max_batch;
    c->scount.assign(c->G, 0); c->sdisp.assign(c->G, 0); c->rcount.assign(c->G, 0); c->rdisp.assign(c->G, 0);
    int rc = mee_router_create(v.device, max_batch, c->G, &c->router);
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, uint64_t bytes) { if (rc == MEE_OK && e == hipSuccess) e = hipMalloc(p, bytes); };
    const uint64_t nslot = c->G > 64 ? c->G : 64;
    alloc((void**)&c->send_keys, max_batch * 8); alloc((void**)&c->perm, max_batch * 8);
    alloc((void**)&c->counts, c->G * 8); alloc((void**)&c->d_recv_counts, nslot * 8);
    alloc((void**)&c->back_rows, c->send_slots * (uint64_t)c->dim * 4); alloc((void**)&c->back_found, c->send_slots);
    alloc((void**)&c->status, 4);
    if (c->cap) alloc((void**)&c->pad_keys, c->send_slots * 8);
    alloc((void**)&c->recv_keys, c->recv_slots * 8); alloc((void**)&c->recv_rows, c->recv_slots * (uint64_t)c->dim * 4);
    alloc((void**)&c->recv_found, c->recv_slots);
    if (c->cold) alloc((void**)&c->recv_found2, c->recv_slots);
    if (c->dedup) {
        alloc((void**)&c->uniq, max_batch * 8); alloc((void**)&c->inverse, max_batch * 8);
        alloc((void**)&c->urows, (max_batch + 1) * (uint64_t)c->dim * 4); alloc((void**)&c->ufound, max_batch + 1);
    }
    if (rc == MEE_OK && e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, 2 * nslot * 8);
    if (rc == MEE_OK && e == hipSuccess) e = hipMemset(c->status, 0, 4);
    if (rc == MEE_OK && e != hipSuccess) rc = fail(MEE_ERR_OUT_OF_MEMORY, "mee_sharded_create_ex: %s", hipGetErrorString(e));
    if (rc == MEE_OK && c->dedup) {   // grouping scratch of our own: a 16-slot table whose group table / per-position arrays serve batches of max_batch keys
        mee_config dc;
        memset(&dc, 0, sizeof dc);
        dc.struct_size = sizeof dc; dc.device = v.device; dc.capacity = 16; dc.dim = v.dim /* mee_dedup_sum adds up rows of the table's dim */; dc.optimizer = MEE_OPT_NONE; dc.max_batch = max_batch;
        rc = mee_table_create(&dc, &c->dd);
        if (rc == MEE_OK) {
            shard_fill_row_kernel<<<1, 256, 0, 0>>>(c->urows + max_batch * (uint64_t)c->dim, c->dim, c->defv, c->ufound + max_batch);
            if (hipDeviceSynchronize() != hipSuccess) rc = fail(MEE_ERR_HIP, "mee_sharded_create_ex: initialising the dedup buffers failed");
        }
    }
    if (rc == MEE_OK && c->cap)   // padded layout: constant message sizes
        for (uint32_t p = 0; p < c->G; ++p) { c->scount[p] = c->rcount[p] = (size_t)c->cap; c->sdisp[p] = c->rdisp[p] = (size_t)p * c->cap; }
    // collective check: owner-side buffers are sized for G x max_batch, so every rank must have passed the same max_batch (and layout).  Each
    // rank writes its value into its own slot of a zeroed vector, one all-reduce (sum) gives everybody all values.
    if (rc == MEE_OK && c->G > 1) {
        for (uint32_t p = 0; p < c->G; ++p) c->h_counts[p] = 0;
        c->h_counts[c->rank] = max_batch * 2 + (c->cap ? 1 : 0) + ((uint64_t)c->cap << 32);
        hipError_t he = hipMemcpy(c->d_recv_counts, c->h_counts, c->G * 8, hipMemcpyHostToDevice);
        ncclResult_t ne = he == hipSuccess ? api->AllReduce(c->d_recv_counts, c->d_recv_counts, c->G, ncclUint64, ncclSum, comm, 0) : ncclSuccess;
        if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpy(c->h_counts, c->d_recv_counts, c->G * 8, hipMemcpyDeviceToHost);
        if (he != hipSuccess) rc = fail(MEE_ERR_HIP, "mee_sharded_create_ex: %s", hipGetErrorString(he));
        else if (ne != ncclSuccess) rc = fail(MEE_ERR_RCCL, "mee_sharded_create_ex: ncclAllReduce failed: %s", api->GetErrorString(ne));
        else
            for (uint32_t p = 0; p < c->G; ++p)
                if (c->h_counts[p] != c->h_counts[c->rank]) {
                    rc = fail(MEE_ERR_INVALID_ARG, "mee_sharded_create_ex: rank %u was created with another max_batch / pad_slack than rank %u (all ranks must agree)", p, c->rank);
                    break;
                }
    }
    if (rc != MEE_OK) { sharded_free(c); delete c; return rc; }
    *out = c;
    return MEE_OK;
}

int mee_sharded_create(mee_table* local, void* nccl_comm, uint64_t max_batch, double pad_slack, mee_sharded** out) {
    MEE_RANGE("mee_sharded_create");
    mee_sharded_options o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o; o.max_batch = max_batch; o.pad_slack = pad_slack;
    return mee_sharded_create_ex(local, nccl_comm, &o, out);
}

static int sharded_lookup(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream, bool insert_missing,
                          const char* name) {
    if (int rc = check_call(c, n, name)) return rc;
    if (n && (!d_keys || !d_out)) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (!c->dedup) {
        if (int rc = push(c, d_keys, nullptr, n, st, &rt)) return rc;   // collective even when n == 0: peers may have keys for this shard
        if (int rc = owner_lookup(c, rt, insert_missing, stream)) return rc;
        return give_back(c, true, d_out, d_found, st);
    }
    // pre-exchange dedup: only the batch's DISTINCT keys travel (keys out, rows back); uniq = the distinct keys followed by EMPTY padding
    // (which belongs to no shard), inverse[i] = the index of position i's key in uniq (max_batch for reserved keys: the default row)
    if (n) if (int rc = mee_dedup_keys(c->dd, d_keys, n, c->uniq, c->inverse, (int64_t)c->max_batch, stream)) return rc;
    if (int rc = push(c, c->uniq, nullptr, n, st, &rt, /*skip_padding=*/true)) return rc;
    if (int rc = owner_lookup(c, rt, insert_missing, stream)) return rc;
    if (int rc = give_back(c, true, c->urows, c->ufound, st)) return rc;
    if (n) {
        shard_expand_kernel<<<grid_for(n, 64, 1u << 16), 256, 0, st>>>((const float4*)c->urows, c->ufound, c->inverse, n, c->dim4, (float4*)d_out, d_found);
        MEE_HIP(hipGetLastError());
    }
    return MEE_OK;
}

int mee_sharded_find(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_sharded_find");
    return sharded_lookup(c, d_keys, n, d_out, d_found, stream, false, "mee_sharded_find");
}
int mee_sharded_find_or_insert(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_sharded_find_or_insert");
    return sharded_lookup(c, d_keys, n, d_out, d_found, stream, true, "mee_sharded_find_or_insert");
}

int mee_sharded_insert(mee_sharded* c, const int64_t* d_keys, const float* d_values, size_t n, void* stream) {
    MEE_RANGE("mee_sharded_insert");
    if (int rc = check_call(c, n, "mee_sharded_insert")) return rc;
    if (n && (!d_keys || !d_values)) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_insert: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = ensure_send_rows(c)) return rc;
    if (int rc = push(c, d_keys, d_values ? d_values : c->send_rows, n, st, &rt)) return rc;
    return owner_insert(c, rt, stream);
}

int mee_sharded_assign(mee_sharded* c, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_sharded_assign");
    if (int rc = check_call(c, n, "mee_sharded_assign")) return rc;
    if (n && (!d_keys || !d_values)) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_assign: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = ensure_send_rows(c)) return rc;
    if (int rc = push(c, d_keys, d_values ? d_values : c->send_rows, n, st, &rt)) return rc;
    if (int rc = owner_assign_or_remove(c, rt, true, stream)) return rc;
    return give_back(c, false, nullptr, d_found, st);
}

int mee_sharded_remove(mee_sharded* c, const int64_t* d_keys, size_t n, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_sharded_remove");
    if (int rc = check_call(c, n, "mee_sharded_remove")) return rc;
    if (n && !d_keys) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_remove: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = push(c, d_keys, nullptr, n, st, &rt)) return rc;
    if (int rc = owner_assign_or_remove(c, rt, false, stream)) return rc;
    return give_back(c, false, nullptr, d_found, st);
}

// one update per distinct key over everything that arrived
static int sharded_apply(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, void* stream, const char* name, const ApplySpec& a) {
    if (int rc = check_call(c, n, name)) return rc;
    if (n && (!d_keys || !d_grads)) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    // padded layout: what arrives is a constant, G x segment capacity — the same on every rank, so a table that is too small is refused
    // on every rank alike, BEFORE anything is exchanged (nobody is left waiting in a collective)
    if (c->cap && (uint64_t)c->G * c->cap > c->local_max_batch)
        return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: the padded layout hands the local table G x segment = %llu positions per apply; it was created with max_batch=%llu",
                    name, (unsigned long long)((uint64_t)c->G * c->cap), (unsigned long long)c->local_max_batch);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = ensure_send_rows(c)) return rc;
    if (c->dedup) {
        // pre-exchange aggregation: the rank's gradient rows of one key are added up (fp64, rounded once) BEFORE they travel — one (key, row) pair per distinct
        // key of the batch crosses xGMI.  The owner's apply then sums the ranks' partial sums in fp64 again: the update differs from the un-aggregated one by
        // the one extra rounding of each rank's partial sum (<= 1 ulp of the partial sum, far inside SPEC.md §4's 1e-6).  Sync-free: the padded unique list
        // goes through mee_partition_padded, which gives the padding to no shard.
        if (n) if (int rc = mee_dedup_sum(c->dd, d_keys, d_grads, n, c->uniq, c->urows, nullptr, nullptr, 0, stream)) return rc;
        if (int rc = push(c, c->uniq, c->urows, n, st, &rt, /*skip_padding=*/true)) return rc;
    } else if (int rc = push(c, d_keys, d_grads ? d_grads : c->send_rows, n, st, &rt)) return rc;
    if (rt == 0) return MEE_OK;
    // (from here on no collective step is left in this operator: a local failure cannot strand a peer)
    if (rt <= c->local_max_batch) return apply_tiers(c, c->recv_keys, c->recv_rows, rt, a, stream);
    return chunked_apply(c, rt, a, stream, name);
}
int mee_sharded_apply_adagrad(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps, void* stream) {
    MEE_RANGE("mee_sharded_apply_adagrad");
    ApplySpec a{false, lr, 0.f, 0.f, eps, 0};
    return sharded_apply(c, d_keys, d_grads, n, stream, "mee_sharded_apply_adagrad", a);
}
int mee_sharded_apply_adam(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1, float beta2, float eps,
                           uint64_t step, void* stream) {
    MEE_RANGE("mee_sharded_apply_adam");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_apply_adam: step must be >= 1");
    ApplySpec a{true, lr, beta1, beta2, eps, step};
    return sharded_apply(c, d_keys, d_grads, n, stream, "mee_sharded_apply_adam", a);
}

int mee_sharded_size(mee_sharded* c, size_t* n_out, void* stream) {
    MEE_RANGE("mee_sharded_size");
    if (!c || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_size: null argument");
    if (c->dead || comm_is_aborted((void*)c->comm)) return fail(MEE_ERR_RCCL, "mee_sharded_size: an earlier RCCL error aborted this context's communicator");
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    size_t mine = 0, cold = 0;
    if (int rc = mee_size(c->local, &mine, stream)) return rc;
    if (c->cold) if (int rc = mee_size(c->cold, &cold, stream)) return rc;
    c->hot_ub = mine;
    c->h_counts[0] = (uint64_t)(mine + cold);
    MEE_HIP(hipMemcpyAsync(c->d_recv_counts, c->h_counts, 8, hipMemcpyHostToDevice, st));
    if (c->G > 1) {
        const ncclResult_t r = api->AllReduce(c->d_recv_counts, c->d_recv_counts, 1, ncclUint64, ncclSum, c->comm, st);
        if (r != ncclSuccess) return rccl_failed(c, api, r, "ncclAllReduce");
    }
    MEE_HIP(hipMemcpyAsync(c->h_counts, c->d_recv_counts, 8, hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)c->h_counts[0];
    return MEE_OK;
}

int mee_sharded_status(mee_sharded* c, uint32_t* bits_out, void* stream) {
    MEE_RANGE("mee_sharded_status");
    if (!c || !bits_out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_status: null argument");
    DeviceGuard g(c->device);
    MEE_HIP(hipMemcpyAsync(bits_out, c->status, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MEE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MEE_OK;
}

int mee_sharded_clear_status(mee_sharded* c, void* stream) {
    MEE_RANGE("mee_sharded_clear_status");
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_clear_status: null context");
    DeviceGuard g(c->device);
    MEE_HIP(hipMemsetAsync(c->status, 0, 4, (hipStream_t)stream));
    return MEE_OK;
}

int mee_sharded_info(const mee_sharded* c, uint32_t* n_shards, uint32_t* rank, uint64_t* segment_capacity) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_info: null context");
    if (n_shards) *n_shards = c->G;
    if (rank) *rank = c->rank;
    if (segment_capacity) *segment_capacity = c->cap;
    return MEE_OK;
}

}  // extern "C"

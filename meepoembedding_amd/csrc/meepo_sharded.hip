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
__global__ __launch_bounds__(256) void shard_return_kernel(const float4* __restrict__ back_rows, const uint8_t* __restrict__ back_found,
                                                           const int64_t* __restrict__ perm, const uint64_t* __restrict__ counts,
                                                           uint64_t cap /* 0 = exact */, uint32_t dim4, float4* __restrict__ out,
                                                           uint8_t* __restrict__ found) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t p = blockIdx.y;
    const uint64_t cnt = counts[p], take = (cap && cnt > cap) ? cap : cnt, b0 = seg_base(counts, p);
    const uint64_t off = cap ? (uint64_t)p * cap : b0;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t j0 = wave * 4; j0 < take; j0 += n_waves * 4) {
        const uint64_t j = j0 + tile;
        if (j >= take) continue;
        const uint64_t dstp = (uint64_t)perm[b0 + j];
        if (back_rows && out)
            for (uint32_t c = tl; c < dim4; c += 16) out[dstp * dim4 + c] = back_rows[(off + j) * dim4 + c];
        if (back_found && found && tl == 0) found[dstp] = back_found[off + j];
    }
}

}  // namespace mee

using namespace mee;

struct mee_sharded {
    int device;
    uint32_t G, rank, dim, dim4;
    uint64_t max_batch;       // lookups / pairs per call of THIS rank
    uint64_t cap;             // padded layout: positions per (source, owner) segment; 0 = exact layout
    uint64_t local_max_batch; // the local table's config.max_batch (bounds one local mutator call)
    uint32_t optimizer;
    mee_table* local;
    mee_router* router;
    ncclComm_t comm;
    // requester side
    int64_t *send_keys, *perm, *pad_keys;
    uint64_t* counts;          // [G] device: keys this rank sends to each owner
    uint64_t* d_recv_counts;   // [G] device: keys each source sends here (exact layout)
    uint64_t* h_counts;        // [2G] pinned: counts, recv counts
    float* send_rows;          // payload rows in send order (allocated by the first mutator call)
    float* back_rows;          // rows returned by the owners, in send order
    uint8_t* back_found;
    uint64_t send_slots;       // positions the requester-side buffers hold: max_batch (exact) or G*cap (padded)
    // owner side
    uint64_t recv_slots;       // positions the owner-side buffers hold (exact: grows on demand)
    int64_t* recv_keys;
    float* recv_rows;          // payload received / rows found
    uint8_t* recv_found;
    uint32_t* status;          // device: bit 0 = a padded segment overflowed
    std::vector<size_t> scount, sdisp, rcount, rdisp;   // in positions
};

namespace mee {

static void sharded_free(mee_sharded* c) {
    void* dev[] = {c->send_keys, c->perm, c->pad_keys, c->counts, c->d_recv_counts, c->send_rows, c->back_rows, c->back_found,
                   c->recv_keys, c->recv_rows, c->recv_found, c->status};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->router) mee_router_destroy(c->router);
}

// owner-side buffers for `slots` positions (exact layout: called whenever a batch brings more than the buffers hold)
static int ensure_recv(mee_sharded* c, uint64_t slots) {
    if (slots <= c->recv_slots && c->recv_keys) return MEE_OK;
    const uint64_t want = slots + slots / 4 + 1024;
    (void)hipDeviceSynchronize();   // rare: the old buffers may still be in use by queued work
    void* old[] = {c->recv_keys, c->recv_rows, c->recv_found};
    for (void* p : old) if (p) (void)hipFree(p);
    c->recv_keys = nullptr; c->recv_rows = nullptr; c->recv_found = nullptr; c->recv_slots = 0;
    MEE_HIP(hipMalloc((void**)&c->recv_keys, want * 8));
    MEE_HIP(hipMalloc((void**)&c->recv_rows, want * (uint64_t)c->dim * 4));
    MEE_HIP(hipMalloc((void**)&c->recv_found, want));
    c->recv_slots = want;
    return MEE_OK;
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
    MEE_NCCL(api, api->GroupStart());
    ncclResult_t first_err = ncclSuccess;   // an open group must be closed whatever happens inside it
    for (uint32_t p = 0; p < c->G && first_err == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        for (int l = 0; l < n_legs && first_err == ncclSuccess; ++l) {
            const size_t row = legs[l].elems * legs[l].elem_bytes;
            // zero-length messages are skipped on both sides (the two ends agree on every count)
            if (sc[p]) first_err = api->Send((const char*)legs[l].send + sd[p] * row, sc[p] * legs[l].elems, legs[l].dt, (int)p, c->comm, st);
            if (rc[p] && first_err == ncclSuccess) first_err = api->Recv((char*)legs[l].recv + rd[p] * row, rc[p] * legs[l].elems, legs[l].dt, (int)p, c->comm, st);
        }
    }
    const ncclResult_t end_err = api->GroupEnd();
    if (first_err != ncclSuccess) return fail(MEE_ERR_RCCL, "ncclSend/ncclRecv failed: %s", api->GetErrorString(first_err));
    if (end_err != ncclSuccess) return fail(MEE_ERR_RCCL, "ncclGroupEnd failed: %s", api->GetErrorString(end_err));
    return MEE_OK;
}

// partition + (exact) counts exchange and the one host synchronisation | (padded) pad the key segments.
// On return keys_to_send points at the keys in segment layout and *r_total = positions the owner side will hold.
static int route(mee_sharded* c, const int64_t* d_keys, size_t n, hipStream_t st, const int64_t** keys_to_send, uint64_t* r_total) {
    if (int rc = mee_partition(c->router, d_keys, n, c->send_keys, c->counts, c->perm, st)) return rc;
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
        MEE_NCCL(api, api->GroupStart());
        ncclResult_t first_err = ncclSuccess;
        for (uint32_t p = 0; p < c->G && first_err == ncclSuccess; ++p) {
            if (p == c->rank) continue;
            first_err = api->Send(c->counts + p, 1, ncclUint64, (int)p, c->comm, st);
            if (first_err == ncclSuccess) first_err = api->Recv(c->d_recv_counts + p, 1, ncclUint64, (int)p, c->comm, st);
        }
        const ncclResult_t end_err = api->GroupEnd();
        if (first_err != ncclSuccess || end_err != ncclSuccess)
            return fail(MEE_ERR_RCCL, "counts exchange failed: %s", api->GetErrorString(first_err != ncclSuccess ? first_err : end_err));
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
    if (int rc = ensure_recv(c, r_acc)) return rc;
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

static int give_back(mee_sharded* c, bool rows, float* d_out, uint8_t* d_found, hipStream_t st) {
    Leg legs[2];
    int nl = 0;
    if (rows) legs[nl++] = Leg{c->recv_rows, c->back_rows, c->dim, 4, ncclFloat};
    legs[nl++] = Leg{c->recv_found, c->back_found, 1, 1, ncclUint8};
    if (int rc = exchange(c, legs, nl, /*reverse=*/true, st)) return rc;
    const dim3 grid(grid_for(c->max_batch / c->G + 64, 16, 4096), c->G);
    shard_return_kernel<<<grid, 256, 0, st>>>(rows ? (const float4*)c->back_rows : nullptr, c->back_found, c->perm, c->counts, c->cap, c->dim4,
                                              (float4*)d_out, d_found);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

static int check_call(const mee_sharded* c, size_t n, const char* name) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "%s: null context", name);
    if (n > c->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: n=%zu exceeds the context's max_batch=%llu", name, n, (unsigned long long)c->max_batch);
    return MEE_OK;
}

// keys (+ payload rows) to their owners; on return recv_keys / recv_rows [0, *r_total) hold what arrived
static int push(mee_sharded* c, const int64_t* d_keys, const float* d_rows, size_t n, hipStream_t st, uint64_t* r_total) {
    const int64_t* ks = nullptr;
    if (int rc = route(c, d_keys, n, st, &ks, r_total)) return rc;
    Leg legs[2];
    int nl = 0;
    legs[nl++] = Leg{ks, c->recv_keys, 1, 8, ncclInt64};
    if (d_rows) {
        if (int rc = pack_rows(c, d_rows, st)) return rc;
        legs[nl++] = Leg{c->send_rows, c->recv_rows, c->dim, 4, ncclFloat};
    }
    return exchange(c, legs, nl, /*reverse=*/false, st);
}

}  // namespace mee

extern "C" {

int mee_comm_unique_id(void* id_out) {
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
    *comm_out = comm;
    return MEE_OK;
}

int mee_comm_destroy(void* comm) {
    if (!comm) return MEE_OK;
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    MEE_NCCL(api, api->CommDestroy((ncclComm_t)comm));
    return MEE_OK;
}

int mee_sharded_destroy(mee_sharded* c) {
    if (!c) return MEE_OK;
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    sharded_free(c);
    delete c;
    return MEE_OK;
}

int mee_sharded_create(mee_table* local, void* nccl_comm, uint64_t max_batch, double pad_slack, mee_sharded** out) {
    if (!out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create: null out");
    *out = nullptr;
    if (!local || !nccl_comm || max_batch == 0 || max_batch > (1ull << 30) || pad_slack < 0.0 || (pad_slack > 0.0 && pad_slack < 1.0))
        return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create: null table / communicator, max_batch not in [1, 2^30], or pad_slack not 0 or >= 1");
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    int nranks = 0, rank = 0, cdev = -1;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    MEE_NCCL(api, api->CommCount(comm, &nranks));
    MEE_NCCL(api, api->CommUserRank(comm, &rank));
    MEE_NCCL(api, api->CommCuDevice(comm, &cdev));
    const TableView v = table_view(local);
    if (cdev != v.device) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create: communicator lives on device %d, the table on device %d", cdev, v.device);
    if (nranks < 1 || nranks > 64) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_create: %d ranks (1..64 supported)", nranks);
    mee_table_info info;
    if (int rc = mee_table_info_get(local, &info)) return rc;
    DeviceGuard g(v.device);
    mee_sharded* c = new (std::nothrow) mee_sharded();
    if (!c) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = v.device; c->G = (uint32_t)nranks; c->rank = (uint32_t)rank; c->dim = v.dim; c->dim4 = v.dim4;
    c->max_batch = max_batch; c->local = local; c->comm = comm; c->local_max_batch = info.max_batch; c->optimizer = info.optimizer;
    c->cap = pad_slack > 0.0 ? (uint64_t)std::ceil((double)max_batch / nranks * pad_slack) + 1024 : 0;
    c->send_slots = c->cap ? (uint64_t)c->G * c->cap : max_batch;
    c->scount.assign(c->G, 0); c->sdisp.assign(c->G, 0); c->rcount.assign(c->G, 0); c->rdisp.assign(c->G, 0);
    int rc = mee_router_create(v.device, max_batch, c->G, &c->router);
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, uint64_t bytes) { if (rc == MEE_OK && e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void**)&c->send_keys, max_batch * 8); alloc((void**)&c->perm, max_batch * 8);
    alloc((void**)&c->counts, c->G * 8); alloc((void**)&c->d_recv_counts, c->G * 8);
    alloc((void**)&c->back_rows, c->send_slots * (uint64_t)c->dim * 4); alloc((void**)&c->back_found, c->send_slots);
    alloc((void**)&c->status, 4);
    if (c->cap) alloc((void**)&c->pad_keys, c->send_slots * 8);
    if (rc == MEE_OK && e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, 2 * c->G * 8);
    if (rc == MEE_OK && e == hipSuccess) e = hipMemset(c->status, 0, 4);
    if (rc == MEE_OK && e != hipSuccess) rc = fail(MEE_ERR_OUT_OF_MEMORY, "mee_sharded_create: %s", hipGetErrorString(e));
    if (rc == MEE_OK) {
        if (c->cap) {   // padded layout: constant message sizes
            for (uint32_t p = 0; p < c->G; ++p) { c->scount[p] = c->rcount[p] = (size_t)c->cap; c->sdisp[p] = c->rdisp[p] = (size_t)p * c->cap; }
            rc = ensure_recv(c, (uint64_t)c->G * c->cap);
        } else {
            rc = ensure_recv(c, max_batch);
        }
    }
    if (rc != MEE_OK) { sharded_free(c); delete c; return rc; }
    *out = c;
    return MEE_OK;
}

static int sharded_lookup(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream, bool insert_missing,
                          const char* name) {
    if (int rc = check_call(c, n, name)) return rc;
    if (n && (!d_keys || !d_out)) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = push(c, d_keys, nullptr, n, st, &rt)) return rc;   // collective even when n == 0: peers may have keys for this shard
    if (insert_missing) {
        for (uint64_t s = 0; s < rt; s += c->local_max_batch) {   // find_or_insert is sequentially consistent: chunks are fine
            const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
            if (int rc = mee_find_or_insert(c->local, c->recv_keys + s, m, c->recv_rows + s * c->dim, c->recv_found + s, stream)) return rc;
        }
    } else if (rt) {
        if (int rc = mee_find(c->local, c->recv_keys, rt, c->recv_rows, c->recv_found, stream)) return rc;
    }
    return give_back(c, true, d_out, d_found, st);
}

int mee_sharded_find(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    return sharded_lookup(c, d_keys, n, d_out, d_found, stream, false, "mee_sharded_find");
}
int mee_sharded_find_or_insert(mee_sharded* c, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    return sharded_lookup(c, d_keys, n, d_out, d_found, stream, true, "mee_sharded_find_or_insert");
}

int mee_sharded_insert(mee_sharded* c, const int64_t* d_keys, const float* d_values, size_t n, void* stream) {
    if (int rc = check_call(c, n, "mee_sharded_insert")) return rc;
    if (n && (!d_keys || !d_values)) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_insert: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = ensure_send_rows(c)) return rc;
    if (int rc = push(c, d_keys, d_values ? d_values : c->send_rows, n, st, &rt)) return rc;
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {   // upserts are sequentially consistent: order kept = last-wins kept
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        if (int rc = mee_insert(c->local, c->recv_keys + s, c->recv_rows + s * c->dim, m, stream)) return rc;
    }
    return MEE_OK;
}

int mee_sharded_assign(mee_sharded* c, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream) {
    if (int rc = check_call(c, n, "mee_sharded_assign")) return rc;
    if (n && (!d_keys || !d_values)) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_assign: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = ensure_send_rows(c)) return rc;
    if (int rc = push(c, d_keys, d_values ? d_values : c->send_rows, n, st, &rt)) return rc;
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        if (int rc = mee_assign(c->local, c->recv_keys + s, c->recv_rows + s * c->dim, m, c->recv_found + s, stream)) return rc;
    }
    return give_back(c, false, nullptr, d_found, st);
}

int mee_sharded_remove(mee_sharded* c, const int64_t* d_keys, size_t n, uint8_t* d_found, void* stream) {
    if (int rc = check_call(c, n, "mee_sharded_remove")) return rc;
    if (n && !d_keys) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_remove: null argument");
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    uint64_t rt = 0;
    if (int rc = push(c, d_keys, nullptr, n, st, &rt)) return rc;
    for (uint64_t s = 0; s < rt; s += c->local_max_batch) {
        const uint64_t m = rt - s < c->local_max_batch ? rt - s : c->local_max_batch;
        if (int rc = mee_remove(c->local, c->recv_keys + s, m, c->recv_found + s, stream)) return rc;
    }
    return give_back(c, false, nullptr, d_found, st);
}

// one update per distinct key over everything that arrived: cannot be chunked, so the local table's max_batch must cover it
static int sharded_apply(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, void* stream, const char* name, uint64_t* rt_out) {
    if (int rc = check_call(c, n, name)) return rc;
    if (n && (!d_keys || !d_grads)) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ensure_send_rows(c)) return rc;
    if (int rc = push(c, d_keys, d_grads ? d_grads : c->send_rows, n, st, rt_out)) return rc;
    if (*rt_out > c->local_max_batch)   // the exchange is complete on every rank: failing here cannot strand a peer in a collective
        return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: %llu pairs arrived at this shard but its table was created with max_batch=%llu", name,
                    (unsigned long long)*rt_out, (unsigned long long)c->local_max_batch);
    return MEE_OK;
}
int mee_sharded_apply_adagrad(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps, void* stream) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_apply_adagrad: null context");
    DeviceGuard g(c->device);
    uint64_t rt = 0;
    if (int rc = sharded_apply(c, d_keys, d_grads, n, stream, "mee_sharded_apply_adagrad", &rt)) return rc;
    return rt ? mee_apply_adagrad(c->local, c->recv_keys, c->recv_rows, rt, lr, eps, stream) : MEE_OK;
}
int mee_sharded_apply_adam(mee_sharded* c, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1, float beta2, float eps,
                           uint64_t step, void* stream) {
    if (!c) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_apply_adam: null context");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_apply_adam: step must be >= 1");
    DeviceGuard g(c->device);
    uint64_t rt = 0;
    if (int rc = sharded_apply(c, d_keys, d_grads, n, stream, "mee_sharded_apply_adam", &rt)) return rc;
    return rt ? mee_apply_adam(c->local, c->recv_keys, c->recv_rows, rt, lr, beta1, beta2, eps, step, stream) : MEE_OK;
}

int mee_sharded_size(mee_sharded* c, size_t* n_out, void* stream) {
    if (!c || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_size: null argument");
    RcclApi* api = rccl_api();
    if (!api) return fail(MEE_ERR_RCCL, "%s", rccl_load_error());
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)stream;
    size_t mine = 0;
    if (int rc = mee_size(c->local, &mine, stream)) return rc;
    c->h_counts[0] = (uint64_t)mine;
    MEE_HIP(hipMemcpyAsync(c->d_recv_counts, c->h_counts, 8, hipMemcpyHostToDevice, st));
    if (c->G > 1) MEE_NCCL(api, api->AllReduce(c->d_recv_counts, c->d_recv_counts, 1, ncclUint64, ncclSum, c->comm, st));
    MEE_HIP(hipMemcpyAsync(c->h_counts, c->d_recv_counts, 8, hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)c->h_counts[0];
    return MEE_OK;
}

int mee_sharded_status(mee_sharded* c, uint32_t* bits_out, void* stream) {
    if (!c || !bits_out) return fail(MEE_ERR_INVALID_ARG, "mee_sharded_status: null argument");
    DeviceGuard g(c->device);
    MEE_HIP(hipMemcpyAsync(bits_out, c->status, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MEE_HIP(hipStreamSynchronize((hipStream_t)stream));
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

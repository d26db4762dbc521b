// meepo_group.hip — one find launch over the batches of MANY tables (a model's embedding collection).
//
// A recommendation model looks up tens of tables per step, each with a small batch; a find launch has a latency floor
// of ~7.6 us (dispatch + three dependent memory round trips, profiles/r01_find.md), so T separate launches cost T floors.
// A group holds the tables' plane descriptors on the device; mee_find_grouped serves the concatenated key batch
// (segment j = the keys of table j, segment bounds in a DEVICE offsets array, "jagged" layout) with ONE kernel whose
// tiles first locate their segment (binary search over the offsets, staged in LDS) and then run the ordinary probe +
// row gather against that table's planes.  Results are identical to calling mee_find per table (SPEC.md §3).
//
// Reference anchor: /root/reference/README.md:2 ("lookuptable-style … Embedding designed for recommendation systems").
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "meepo_device.h"
#include "meepo_host.h"

namespace mee {

typedef float f32x4 __attribute__((ext_vector_type(4)));


// One tile per key, R keys in flight per tile (same shape as find_kernel).  DIM4 = dim/4 when it is 16 or 32, 0 = any.
// LOCATE: no rows move; gslot[i] = member << 48 | slot of the key (kEmpty when absent / reserved / outside the segments) —
// the first pass of a grouped apply.
template <int DIM4, int R, bool STREAM_OUT, bool LOCATE = false>
__global__ __launch_bounds__(256) void find_grouped_kernel(const GroupDesc* __restrict__ desc, uint32_t n_tables,
                                                           const uint64_t* __restrict__ offsets, const int64_t* __restrict__ keys,
                                                           uint64_t n, float4* __restrict__ out, uint8_t* __restrict__ found,
                                                           uint32_t dim4_rt, int64_t* __restrict__ gslot = nullptr,
                                                           const GroupInit* __restrict__ init = nullptr, uint64_t off_stride = 1) {
    __shared__ uint64_t loff[kMaxGroupTables + 1];
    for (uint32_t j = threadIdx.x; j <= n_tables; j += blockDim.x) loff[j] = offsets[(uint64_t)j * off_stride];  // stride > 1: the
    __syncthreads();                                                     // segment bounds are every stride-th entry of a bag-offset array
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    constexpr int C = DIM4 ? DIM4 / 16 : 1;
    for (uint64_t base = wave * 4 * R; base < n; base += n_waves * 4 * R) {
        int64_t key[R], slot[R], kb[R];
        uint64_t b[R];
        GroupDesc d[R];
        uint32_t member[R];
        bool inb[R], act[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = base + r * 4 + tile;
            // segment of position i: the last j with loff[j] <= i (empty segments are skipped by the search itself)
            uint32_t lo = 0, hi = n_tables;   // invariant: loff[lo] <= i < loff[hi] once i is inside [loff[0], loff[n_tables])
            inb[r] = i < n && i >= loff[0] && i < loff[n_tables];
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (loff[mid] <= i) lo = mid; else hi = mid;
            }
            member[r] = lo;
            d[r] = desc[inb[r] ? lo : 0];   // 32 B, L1/L2 resident (staging the descriptors in LDS as well measured 3 % slower)
            key[r] = inb[r] ? keys[i] : kEmpty;
            act[r] = inb[r] && !reserved_key(key[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            b[r] = bucket_of(key[r], d[r].nb);
            kb[r] = act[r] ? d[r].tkeys[b[r] * kW + tl] : kEmpty;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            slot[r] = -1;
            bool pend = act[r];
            uint64_t bb = b[r], steps = 0;
            int64_t k = kb[r];
            while (true) {
                const uint32_t tm = tile_bits(__ballot(pend && k == key[r]), tile);
                const uint32_t te = tile_bits(__ballot(pend && k == kEmpty), tile);
                if (pend) {
                    if (tm) { slot[r] = (int64_t)(bb * kW) + (__ffs(tm) - 1); pend = false; }
                    else if (te || ++steps >= d[r].nb) pend = false;
                    else bb = next_bucket(bb, step_of(key[r], d[r].nb), d[r].nb);
                }
                if (!__any(pend)) break;
                k = pend ? d[r].tkeys[bb * kW + tl] : kEmpty;
            }
        }
        if constexpr (LOCATE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (i < n && tl == 0) gslot[i] = slot[r] >= 0 ? (int64_t)(((uint64_t)member[r] << kGroupSlotBits) | (uint64_t)slot[r]) : kEmpty;
                if (inb[r] && key[r] == kReclaimed && tl == 0) atomicOr(init[member[r]].status, (uint32_t)MEE_STATUS_RESERVED_KEY);  // as mee_apply_* does
            }
            continue;
        }
        if constexpr (DIM4 != 0) {
            float4 row[R][C];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c)
                    row[r][c] = slot[r] >= 0 ? d[r].values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]
                                             : make_float4(d[r].defv, d[r].defv, d[r].defv, d[r].defv);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r])
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        if constexpr (STREAM_OUT) {   // a dense output beyond the Infinity Cache: streaming stores (find_kernel's policy)
                            const f32x4 v = {row[r][c].x, row[r][c].y, row[r][c].z, row[r][c].w};
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out) + i * DIM4 + c * 16 + tl);
                        } else {
                            out[i * DIM4 + c * 16 + tl] = row[r][c];
                        }
                    }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r])
                    for (uint32_t c = tl; c < dim4; c += 16)
                        out[i * dim4 + c] = slot[r] >= 0 ? d[r].values[(uint64_t)slot[r] * dim4 + c]
                                                         : make_float4(d[r].defv, d[r].defv, d[r].defv, d[r].defv);
            }
        }
        if (found) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r] && tl == 0) found[i] = slot[r] >= 0;
            }
        }
    }
}

// Second pass of a grouped find_or_insert: every position the find pass left missing claims (or finds, when a duplicate
// got there first) its key's slot in its member table; the tile whose CAS created the key writes the initial row, the
// initial optimizer state and a zero hit counter, and every such tile writes that same initial row — a function of the
// key and its table's initializer alone — into `out`, so nothing has to read the created rows back.  One tile per position.
__global__ __launch_bounds__(256) void ensure_grouped_kernel(const GroupDesc* __restrict__ desc, const GroupInit* __restrict__ init,
                                                             uint32_t n_tables, const uint64_t* __restrict__ offsets,
                                                             const int64_t* __restrict__ keys, uint64_t n,
                                                             const uint8_t* __restrict__ found, uint32_t dim4, float4* __restrict__ out) {
    __shared__ uint64_t loff[kMaxGroupTables + 1];
    for (uint32_t j = threadIdx.x; j <= n_tables; j += blockDim.x) loff[j] = offsets[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint64_t i = base + tile;
        bool act = i < n && i >= loff[0] && i < loff[n_tables] && found[i] == 0;
        const int64_t key = act ? keys[i] : kEmpty;
        const bool tomb = act && key == kReclaimed;   // a reserved key in the batch: flagged like mee_find_or_insert does (EMPTY = padding, silent)
        act = act && !reserved_key(key);
        if (!__any(act || tomb)) continue;  // wave-uniform: nothing missing here (the steady state of a trained vocabulary)
        uint32_t lo = 0, hi = n_tables;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (loff[mid] <= i) lo = mid; else hi = mid;
        }
        if (tomb && tl == 0) atomicOr(init[lo].status, (uint32_t)MEE_STATUS_RESERVED_KEY);
        const GroupDesc d = desc[act ? lo : 0];
        bool is_new, full;
        const int64_t slot = tile_locate<true, true>(const_cast<int64_t*>(d.tkeys), d.nb, key, act, tile, tl, is_new, full);
        if (act) {
            const GroupInit in = init[lo];
            if (slot >= 0) {
                for (uint32_t c = tl; c < dim4; c += 16) {
                    const float4 row = initial_row4(key, c * 4, in.initializer, in.init_scale, in.init_seed, d.defv);
                    out[i * dim4 + c] = row;
                    if (is_new) {
                        d.values[(uint64_t)slot * dim4 + c] = row;
                        if (in.optimizer == MEE_OPT_ADAGRAD) d.s1[(uint64_t)slot * dim4 + c] = make_float4(in.init_acc, in.init_acc, in.init_acc, in.init_acc);
                        if (in.optimizer == MEE_OPT_ADAM) {
                            d.s1[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                            d.s2[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                if (is_new && in.hits && tl == 0) in.hits[slot] = 0;
            }
            if (full && tl == 0) atomicOr(in.status, (uint32_t)MEE_STATUS_TABLE_FULL);
        }
    }
}

}  // namespace mee

using namespace mee;


static int upload_descriptors(mee_group* g) {
    std::vector<GroupDesc> h(g->n_tables);
    std::vector<GroupInit> hi(g->n_tables);
    for (uint32_t j = 0; j < g->n_tables; ++j) {
        const TableView v = table_view(g->tables[j]);
        h[j] = GroupDesc{v.keys, (float4*)v.values, (float4*)v.s1, (float4*)v.s2, v.nb, v.default_value, 0};
        hi[j] = GroupInit{v.init_seed, v.status, v.hits, v.initializer, v.optimizer, v.init_scale, v.init_acc};
        g->generations[j] = v.generation;
    }
    MEE_HIP(hipMemcpy(g->d_desc, h.data(), h.size() * sizeof(GroupDesc), hipMemcpyHostToDevice));  // synchronous, rare
    MEE_HIP(hipMemcpy(g->d_init, hi.data(), hi.size() * sizeof(GroupInit), hipMemcpyHostToDevice));
    return MEE_OK;
}

namespace mee {
int group_refresh(mee_group* g, void* stream) {
    for (uint32_t j = 0; j < g->n_tables; ++j)
        if (table_view(g->tables[j]).generation != g->generations[j]) {   // a table was rehashed: its planes moved
            DeviceGuard guard(g->device);
            MEE_HIP(hipStreamSynchronize((hipStream_t)stream));           // launches in flight may still read the old descriptors
            return upload_descriptors(g);
        }
    return MEE_OK;
}
int group_locate(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, uint64_t off_stride, size_t n, int64_t* d_gslot, hipStream_t st) {
    find_grouped_kernel<0, 2, false, true><<<grid_for(n, 32, 8192), 256, 0, st>>>(g->d_desc, g->n_tables, d_offsets, d_keys, n, nullptr, nullptr, g->dim4, d_gslot, g->d_init, off_stride);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}
}  // namespace mee

extern "C" {

int mee_group_create(mee_table* const* tables, uint32_t n_tables, uint64_t max_apply_batch, mee_group** out) {
    MEE_RANGE("mee_group_create");
    if (!tables || !out || n_tables == 0 || n_tables > kMaxGroupTables)
        return fail(MEE_ERR_INVALID_ARG, "mee_group_create: need 1..%u tables", kMaxGroupTables);
    *out = nullptr;
    for (uint32_t j = 0; j < n_tables; ++j)
        if (!tables[j]) return fail(MEE_ERR_INVALID_ARG, "mee_group_create: table %u is null", j);
    const TableView v0 = table_view(tables[0]);
    for (uint32_t j = 1; j < n_tables; ++j) {
        const TableView v = table_view(tables[j]);
        if (v.device != v0.device || v.dim != v0.dim)
            return fail(MEE_ERR_INVALID_ARG, "mee_group_create: table %u differs from table 0 in device or dim (%d/%u vs %d/%u)", j, v.device, v.dim, v0.device, v0.dim);
        if (max_apply_batch && v.optimizer != v0.optimizer)
            return fail(MEE_ERR_INVALID_ARG, "mee_group_create: a group with grouped apply needs one optimizer for all members (table %u: %u vs %u)", j, v.optimizer, v0.optimizer);
    }
    if (max_apply_batch > (1ull << 30)) return fail(MEE_ERR_INVALID_ARG, "mee_group_create: max_apply_batch must be <= 2^30");
    mee_group* g = new (std::nothrow) mee_group();
    if (!g) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    g->device = v0.device; g->n_tables = n_tables; g->dim = v0.dim; g->dim4 = v0.dim4; g->d_desc = nullptr;
    g->optimizer = v0.optimizer; g->max_apply_batch = max_apply_batch; g->scratch = nullptr; g->d_gslot = nullptr;
    g->d_init = nullptr; g->d_fmask = nullptr;
    g->tables.assign(tables, tables + n_tables);
    g->generations.assign(n_tables, 0);
    DeviceGuard guard(g->device);
    hipError_t e = hipMalloc((void**)&g->d_desc, n_tables * sizeof(GroupDesc));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_init, n_tables * sizeof(GroupInit));
    if (e == hipSuccess && max_apply_batch) e = hipMalloc((void**)&g->d_fmask, max_apply_batch);
    if (e != hipSuccess) { mee_group_destroy(g); return fail(MEE_ERR_OUT_OF_MEMORY, "hipMalloc(group descriptors): %s", hipGetErrorString(e)); }
    if (int rc = upload_descriptors(g)) { mee_group_destroy(g); return rc; }
    if (max_apply_batch && g->optimizer != MEE_OPT_NONE) {
        // the group's own group table, per-position arrays and counters: a 16-slot table that only lends its scratch
        mee_config c{};
        c.struct_size = sizeof c; c.device = g->device; c.capacity = 16; c.dim = g->dim; c.optimizer = g->optimizer; c.max_batch = max_apply_batch;
        int rc = mee_table_create(&c, &g->scratch);
        if (rc == MEE_OK && hipMalloc((void**)&g->d_gslot, max_apply_batch * sizeof(int64_t)) != hipSuccess)
            rc = fail(MEE_ERR_OUT_OF_MEMORY, "hipMalloc(group slot list)");
        if (rc != MEE_OK) { mee_group_destroy(g); return rc; }
    }
    *out = g;
    return MEE_OK;
}

int mee_group_set_tuning(mee_group* g, const char* name, int value) {
    if (!g || !name) return fail(MEE_ERR_INVALID_ARG, "mee_group_set_tuning: null argument");
    if (!g->scratch) return fail(MEE_ERR_UNSUPPORTED, "mee_group_set_tuning: the group has no apply of its own (max_apply_batch = 0 or tables without optimizer)");
    return mee_set_tuning(g->scratch, name, value);
}

int mee_group_destroy(mee_group* g) {
    MEE_RANGE("mee_group_destroy");
    if (!g) return MEE_OK;
    DeviceGuard guard(g->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(g->d_desc);
    (void)hipFree(g->d_init);
    (void)hipFree(g->d_fmask);
    (void)hipFree(g->d_gslot);
    if (g->scratch) mee_table_destroy(g->scratch);
    delete g;
    return MEE_OK;
}

}  // extern "C"

static int launch_find_grouped(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found,
                               hipStream_t st) {
    const bool stream_out = (uint64_t)n * g->dim * 4 > (128ull << 20);   // find_kernel's store policy
    // every block starts by staging the offsets in LDS (a global round trip + a barrier): blocks must live long enough to
    // amortise it, so the grid is capped and strides (measured: 8192 blocks best from 200K to 1M positions)
    const unsigned grid_cap = 8192;
#define GROUPED(D4, RR, PER)                                                                                                          \
    do {                                                                                                                              \
        if (stream_out) find_grouped_kernel<D4, RR, true><<<grid_for(n, PER, grid_cap), 256, 0, st>>>(g->d_desc, g->n_tables, d_offsets, d_keys, n, (float4*)d_out, d_found, g->dim4); \
        else find_grouped_kernel<D4, RR, false><<<grid_for(n, PER, grid_cap), 256, 0, st>>>(g->d_desc, g->n_tables, d_offsets, d_keys, n, (float4*)d_out, d_found, g->dim4);       \
    } while (0)
    if (g->dim4 == 16) GROUPED(16, 2, 32);
    else if (g->dim4 == 32) GROUPED(32, 1, 16);
    else GROUPED(0, 1, 16);
#undef GROUPED
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

extern "C" {

int mee_find_grouped(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found,
                     void* stream) {
    MEE_RANGE("mee_find_grouped");
    if (!g || !d_offsets || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_grouped: null argument");
    if (n == 0) return MEE_OK;
    if (int rc = group_refresh(g, stream)) return rc;
    DeviceGuard guard(g->device);
    return launch_find_grouped(g, d_keys, d_offsets, n, d_out, d_found, (hipStream_t)stream);
}

int mee_group_find_or_insert(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found,
                             void* stream) {
    MEE_RANGE("mee_group_find_or_insert");
    if (!g || !d_offsets || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_group_find_or_insert: null argument");
    if (n == 0) return MEE_OK;
    if (!d_found) {
        if (n > g->max_apply_batch)
            return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_group_find_or_insert: without a found buffer n=%zu must be <= max_apply_batch=%llu", n, (unsigned long long)g->max_apply_batch);
        d_found = g->d_fmask;
    }
    if (int rc = group_refresh(g, stream)) return rc;
    DeviceGuard guard(g->device);
    hipStream_t st = (hipStream_t)stream;
    // 1) the ordinary grouped find serves every stored key and yields the present-before mask
    if (int rc = launch_find_grouped(g, d_keys, d_offsets, n, d_out, d_found, st)) return rc;
    // 2) missing positions create their key (one creator per distinct key: the CAS decides) and return its initial row
    ensure_grouped_kernel<<<grid_for(n, 16, 8192), 256, 0, st>>>(g->d_desc, g->d_init, g->n_tables, d_offsets, d_keys, n, d_found, g->dim4, (float4*)d_out);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

}  // extern "C"

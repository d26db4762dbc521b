// meepo_table.hip — HBM-resident open-addressing table for gfx950: kernels + the table half of the C-ABI.
//
// Reference anchor: /root/reference/README.md:2 ("dynamic lookuptable-style Embedding … Supports GPU …"); the
// snapshot has no code, so semantics come from SPEC.md (§2 table, §3 operators, §4 optimizers).
//
// HBM layout (all slot-indexed, one allocation each):
//   keys   int64[capacity]          16 keys = one 128-B line = one bucket
//   values fp32 [capacity][dim]     row of slot s at values + s*dim  (256 B for dim 64)
//   s1,s2  fp32 [capacity][dim]     optimizer planes (acc | m, v), only if configured
// Per-batch scratch ("group table"): an open-addressing set of the batch's distinct keys sized ≥ 2·max_batch
// (stays in L2 / Infinity Cache), used to give every distinct key one owner tile: duplicate-key reduction
// for the optimizers, last-wins for insert/assign, single insertion for find_or_insert.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "meepo_device.h"
#include "meepo_host.h"

namespace mee {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

struct Counters {            // device-resident, persistent
    unsigned long long size; // stored keys
    uint32_t status;         // sticky MEE_STATUS_* bits
    uint32_t pad;
};
struct OpCounters {          // device-resident, zeroed at the start of each op that uses them
    uint32_t n_uniq;
    uint32_t n_occ;
    unsigned long long n_export;
};

}  // namespace mee

struct mee_table {
    int device;
    uint64_t capacity, nb, max_batch;
    uint32_t dim, dim4, optimizer, initializer;
    float default_value, init_acc, init_scale;
    uint64_t init_seed;
    // table planes
    int64_t* keys;
    float *values, *s1, *s2;
    // group table (scratch), S entries
    uint64_t S, smask;
    unsigned long long* skeys;  // key ^ kBias, 0 = empty
    uint32_t* sval;             // occurrence count | winner index+1
    uint32_t* soffs;            // start of the group's occurrence list
    uint32_t* sgrp;             // index of the group in the unique list
    long long* sres;            // find_or_insert: slot | present<<62, -1 = not stored
    // per batch position
    uint32_t *hidx, *rank, *occ, *uniq_h;
    mee::Counters* ctr;
    mee::OpCounters* op;
    mee::Counters* h_ctr;       // pinned staging for read-backs
    mee::OpCounters* h_op;
    uint64_t table_bytes, workspace_bytes;
    // performance knobs (never change results): see mee_set_tuning()
    int find_rounds;            // keys in flight per tile in the find kernel: 1, 2, 4 or 8
    int find_grid_cap;          // max blocks of the find grid (0 = one pass, no grid-stride loop)
    int find_nt;                // bit0: non-temporal row loads, bit1: non-temporal bucket loads
};

namespace mee {

// =========================================================================================================
// kernels
// =========================================================================================================

__global__ void fill_i64_kernel(int64_t* p, uint64_t n, int64_t v) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- find (SPEC.md §3) — the headline kernel --------------------------------------------------------------
// One tile per key, R keys in flight per tile: the R bucket lines are requested back to back, then the R rows.
// DIM4 = dim/4 when it is a multiple of 16 (each lane moves DIM4/16 float4 per row), 0 = any dim at run time.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIM4, int R, int NT>
__global__ __launch_bounds__(256) void find_kernel(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values,
                                                   uint64_t nb, const int64_t* __restrict__ keys, uint64_t n,
                                                   f32x4* __restrict__ out, uint8_t* __restrict__ found, float defv,
                                                   uint32_t dim4_rt) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    constexpr int KPW = 4 * R;
    const f32x4 def4 = {defv, defv, defv, defv};

    for (uint64_t base = wave * KPW; base < n; base += n_waves * KPW) {
        int64_t key[R];
        int64_t slot[R];
        uint64_t b[R];
        int64_t kb[R];
        bool inb[R], act[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = base + r * 4 + tile;
            inb[r] = i < n;
            key[r] = inb[r] ? keys[i] : kEmpty;
            act[r] = inb[r] && !reserved_key(key[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            b[r] = bucket_of(key[r], nb);
            kb[r] = act[r] ? ((NT & 2) ? __builtin_nontemporal_load(&tkeys[b[r] * kW + tl]) : tkeys[b[r] * kW + tl]) : kEmpty;
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
                    else if (te || ++steps >= nb) pend = false;
                    else bb = (bb + 1 == nb) ? 0 : bb + 1;
                }
                if (!__any(pend)) break;
                k = pend ? tkeys[bb * kW + tl] : kEmpty;
            }
        }
        if constexpr (DIM4 != 0) {
            constexpr int C = DIM4 / 16;
            f32x4 row[R][C];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < C; ++c)
                    row[r][c] = slot[r] >= 0 ? ((NT & 1) ? __builtin_nontemporal_load(&values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]) : values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]) : def4;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r]) {
#pragma unroll
                    for (int c = 0; c < C; ++c) __builtin_nontemporal_store(row[r][c], &out[i * DIM4 + c * 16 + tl]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r])
                    for (uint32_t c = tl; c < dim4; c += 16)
                        out[i * dim4 + c] = slot[r] >= 0 ? values[(uint64_t)slot[r] * dim4 + c] : def4;
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

// ---- group table: one entry per distinct key of the batch ---------------------------------------------------
__device__ __forceinline__ uint32_t group_claim(unsigned long long* skeys, uint64_t smask, int64_t key) {
    const unsigned long long bk = (unsigned long long)key ^ kBias;  // != 0 because key != kEmpty
    uint32_t h = (uint32_t)(mix64b((uint64_t)key) & smask);
    while (true) {
        unsigned long long cur = __hip_atomic_load(&skeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) cur = atomicCAS(&skeys[h], 0ull, bk);
        if (cur == 0 || cur == bk) return h;
        h = (h + 1) & (uint32_t)smask;
    }
}

// COUNT: sval = occurrence count, rank[i] = arrival order inside the group, leaders append to the unique list.
// !COUNT: sval = 1 + highest batch position (last occurrence wins).
template <bool COUNT>
__global__ __launch_bounds__(256) void group_kernel(const int64_t* __restrict__ keys, uint32_t n,
                                                    unsigned long long* skeys, uint64_t smask, uint32_t* sval,
                                                    uint32_t* sgrp, uint32_t* hidx, uint32_t* rank, uint32_t* uniq_h,
                                                    Counters* ctr, OpCounters* op) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool inb = i < n;
    const int64_t key = inb ? keys[i] : 0;
    const bool valid = inb && !reserved_key(key);
    uint32_t h = kNoGroup;
    bool leader = false;
    if (valid) {
        h = group_claim(skeys, smask, key);
        if constexpr (COUNT) {
            const uint32_t r = atomicAdd(&sval[h], 1u);
            rank[i] = r;
            leader = r == 0;
        } else {
            atomicMax(&sval[h], i + 1);
        }
    }
    if (inb) {
        hidx[i] = h;
        if (!valid) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_RESERVED_KEY);
    }
    if constexpr (COUNT) {
        const uint64_t lm = __ballot(leader);
        if (lm) {  // wave-uniform
            const int first = __ffsll((unsigned long long)lm) - 1;
            uint32_t basev = 0;
            if (lane == first) basev = atomicAdd(&op->n_uniq, (uint32_t)__popcll(lm));
            basev = __shfl(basev, first);
            if (leader) {
                const uint32_t u = basev + (uint32_t)__popcll(lm & ((1ull << lane) - 1));
                uniq_h[u] = h;
                sgrp[h] = u;
            }
        }
    }
}

// Allocate each group's slice of the occurrence list: wave prefix-sum + one atomic per wave.
// ALL = every group gets a slice (standalone dedup); otherwise only groups with ≥2 occurrences.
template <bool ALL>
__global__ __launch_bounds__(256) void group_offsets_kernel(const uint32_t* __restrict__ uniq_h,
                                                            const uint32_t* __restrict__ sval, uint32_t* soffs,
                                                            OpCounters* op) {
    const uint32_t nu = op->n_uniq;
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x * blockDim.x >= nu) return;  // block-uniform
    const uint32_t h = u < nu ? uniq_h[u] : 0;
    const uint32_t c = u < nu ? sval[h] : 0;
    const uint32_t need = (ALL || c > 1) ? c : 0;
    uint32_t incl = need;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t basev = 0;
    if (total) {
        if (lane == 63) basev = atomicAdd(&op->n_occ, total);
        basev = __shfl(basev, 63);
    }
    if (u < nu && need) soffs[h] = basev + incl - need;
}

__global__ __launch_bounds__(256) void group_reset_kernel(const uint32_t* __restrict__ hidx, uint32_t n,
                                                          unsigned long long* skeys, uint32_t* sval) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hidx[i];
    if (h != kNoGroup) { skeys[h] = 0; sval[h] = 0; }
}

// ---- insert / assign (SPEC.md §3) --------------------------------------------------------------------------
// Only the winner occurrence (highest batch position) of each distinct key writes.
template <bool CLAIM>
__global__ __launch_bounds__(256) void upsert_kernel(int64_t* tkeys, float4* values, float4* s1, float4* s2, uint64_t nb,
                                                     uint32_t dim4, const int64_t* __restrict__ keys,
                                                     const float4* __restrict__ vals, uint32_t n,
                                                     const uint32_t* __restrict__ hidx, const uint32_t* __restrict__ sval,
                                                     uint8_t* found, uint32_t optimizer, float init_acc, Counters* ctr) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        const bool inb = i < n;
        const int64_t key = inb ? keys[i] : kEmpty;
        const uint32_t h = inb ? hidx[i] : kNoGroup;
        const bool valid = h != kNoGroup;
        const bool winner = valid && sval[h] == i + 1;
        bool is_new, full;
        // insert: only winners touch the table.  assign: every occurrence probes (keys do not change) so that
        // found[] is exact for all of them; only the winner writes.
        const int64_t slot = tile_locate<CLAIM, CLAIM>(tkeys, nb, key, CLAIM ? winner : valid, tile, tl, is_new, full);
        if (winner && slot >= 0) {
            for (uint32_t c = tl; c < dim4; c += 16) {
                values[(uint64_t)slot * dim4 + c] = vals[(uint64_t)i * dim4 + c];
                if (CLAIM && is_new) {
                    if (optimizer == MEE_OPT_ADAGRAD) s1[(uint64_t)slot * dim4 + c] = make_float4(init_acc, init_acc, init_acc, init_acc);
                    if (optimizer == MEE_OPT_ADAM) {
                        s1[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                        s2[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
        if (!CLAIM && found && inb && tl == 0) found[i] = slot >= 0;
        if constexpr (CLAIM) {
            const uint64_t nm = __ballot(is_new && tl == 0);
            const uint64_t fm = __ballot(full);
            if (lane == 0 && nm) atomicAdd(&ctr->size, (unsigned long long)__popcll(nm));
            if (lane == 0 && fm) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_TABLE_FULL);
        }
    }
}

// ---- find_or_insert (SPEC.md §3) ---------------------------------------------------------------------------
constexpr long long kPresentBit = 1ll << 62;

__global__ __launch_bounds__(256) void ensure_kernel(int64_t* tkeys, float4* values, float4* s1, float4* s2, uint64_t nb,
                                                     uint32_t dim4, const int64_t* __restrict__ keys, uint32_t n,
                                                     const uint32_t* __restrict__ hidx, const uint32_t* __restrict__ sval,
                                                     long long* sres, uint32_t optimizer, float init_acc,
                                                     uint32_t initializer, float init_scale, uint64_t init_seed,
                                                     float default_value, Counters* ctr) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        const bool inb = i < n;
        const int64_t key = inb ? keys[i] : kEmpty;
        const uint32_t h = inb ? hidx[i] : kNoGroup;
        const bool winner = h != kNoGroup && sval[h] == i + 1;
        bool is_new, full;
        const int64_t slot = tile_locate<true, true>(tkeys, nb, key, winner, tile, tl, is_new, full);
        if (winner && slot >= 0 && is_new) {
            for (uint32_t c = tl; c < dim4; c += 16) {
                values[(uint64_t)slot * dim4 + c] = initial_row4(key, c * 4, initializer, init_scale, init_seed, default_value);
                if (optimizer == MEE_OPT_ADAGRAD) s1[(uint64_t)slot * dim4 + c] = make_float4(init_acc, init_acc, init_acc, init_acc);
                if (optimizer == MEE_OPT_ADAM) {
                    s1[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    s2[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        if (winner && tl == 0) sres[h] = slot < 0 ? -1ll : ((long long)slot | (is_new ? 0ll : kPresentBit));
        const uint64_t nm = __ballot(is_new && tl == 0);
        const uint64_t fm = __ballot(full);
        if (lane == 0 && nm) atomicAdd(&ctr->size, (unsigned long long)__popcll(nm));
        if (lane == 0 && fm) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_TABLE_FULL);
    }
}

__global__ __launch_bounds__(256) void gather_group_kernel(const float4* __restrict__ values, uint32_t dim4, uint32_t n,
                                                           const uint32_t* __restrict__ hidx,
                                                           const long long* __restrict__ sres, float4* __restrict__ out,
                                                           uint8_t* found, float defv) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const float4 def4 = make_float4(defv, defv, defv, defv);
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        if (i >= n) continue;
        const uint32_t h = hidx[i];
        const long long res = h != kNoGroup ? sres[h] : -1ll;
        const long long slot = res < 0 ? -1ll : (res & ~kPresentBit);
        for (uint32_t c = tl; c < dim4; c += 16) out[(uint64_t)i * dim4 + c] = slot >= 0 ? values[(uint64_t)slot * dim4 + c] : def4;
        if (found && tl == 0) found[i] = res >= 0 && (res & kPresentBit) != 0;
    }
}

// ---- sparse optimizers (SPEC.md §4) ------------------------------------------------------------------------
struct OptArgs {
    uint32_t kind;       // MEE_OPT_*
    float lr, eps;       // adagrad: lr; adam: lr unused (step_size)
    float step_size, omb1, omb2;
};

__device__ __forceinline__ void opt_update4(const OptArgs& a, float4& w, float4& x1, float4& x2, const float4 g) {
    if (a.kind == MEE_OPT_ADAGRAD) {
        adagrad1(w.x, x1.x, g.x, a.lr, a.eps); adagrad1(w.y, x1.y, g.y, a.lr, a.eps);
        adagrad1(w.z, x1.z, g.z, a.lr, a.eps); adagrad1(w.w, x1.w, g.w, a.lr, a.eps);
    } else {
        adam1(w.x, x1.x, x2.x, g.x, a.step_size, a.omb1, a.omb2, a.eps);
        adam1(w.y, x1.y, x2.y, g.y, a.step_size, a.omb1, a.omb2, a.eps);
        adam1(w.z, x1.z, x2.z, g.z, a.step_size, a.omb1, a.omb2, a.eps);
        adam1(w.w, x1.w, x2.w, g.w, a.step_size, a.omb1, a.omb2, a.eps);
    }
}

// Pass 1 over batch positions: a key that occurs once is updated right here from its own grad row (the common
// case); occurrences of multi-keys are filed into their group's occurrence list for pass 2.
__global__ __launch_bounds__(256) void apply_single_kernel(const int64_t* __restrict__ tkeys, float4* values, float4* s1,
                                                           float4* s2, uint64_t nb, uint32_t dim4,
                                                           const int64_t* __restrict__ keys,
                                                           const float4* __restrict__ grads, uint32_t n,
                                                           const uint32_t* __restrict__ hidx,
                                                           const uint32_t* __restrict__ rank,
                                                           const uint32_t* __restrict__ sval,
                                                           const uint32_t* __restrict__ soffs, uint32_t* occ, OptArgs a) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        const bool inb = i < n;
        const int64_t key = inb ? keys[i] : kEmpty;
        const uint32_t h = inb ? hidx[i] : kNoGroup;
        const uint32_t cnt = h != kNoGroup ? sval[h] : 0;
        bool is_new, full;
        const int64_t slot = tile_locate<false, false>(const_cast<int64_t*>(tkeys), nb, key, cnt == 1, tile, tl, is_new, full);
        if (cnt == 1 && slot >= 0) {
            for (uint32_t c = tl; c < dim4; c += 16) {
                const uint64_t o = (uint64_t)slot * dim4 + c;
                const float4 g = grads[(uint64_t)i * dim4 + c];
                float4 w = values[o], x1 = s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.kind == MEE_OPT_ADAM) x2 = s2[o];
                opt_update4(a, w, x1, x2, g);
                values[o] = w; s1[o] = x1;
                if (a.kind == MEE_OPT_ADAM) s2[o] = x2;
            }
        } else if (cnt > 1 && tl == 0) {
            occ[soffs[h] + rank[i]] = i;
        }
    }
}

// Pass 2 over distinct keys: sum the occurrence list of each multi-key in fp64, round once, update once.
// Also returns every group-table entry to the empty state (the table must be clean for the next op).
__global__ __launch_bounds__(256) void apply_multi_kernel(const int64_t* __restrict__ tkeys, float4* values, float4* s1,
                                                          float4* s2, uint64_t nb, uint32_t dim4,
                                                          const float4* __restrict__ grads,
                                                          const uint32_t* __restrict__ uniq_h, unsigned long long* skeys,
                                                          uint32_t* sval, const uint32_t* __restrict__ soffs,
                                                          const uint32_t* __restrict__ occ, const OpCounters* op, OptArgs a) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t nu = op->n_uniq;
    for (uint32_t base = wave * 4; base < nu; base += n_waves * 4) {
        const uint32_t u = base + tile;
        const bool inb = u < nu;
        const uint32_t h = inb ? uniq_h[u] : 0;
        const uint32_t cnt = inb ? sval[h] : 0;
        const int64_t key = inb ? (int64_t)(skeys[h] ^ kBias) : kEmpty;
        bool is_new, full;
        const int64_t slot = tile_locate<false, false>(const_cast<int64_t*>(tkeys), nb, key, cnt > 1, tile, tl, is_new, full);
        if (cnt > 1 && slot >= 0) {
            const uint32_t off = soffs[h];
            for (uint32_t c = tl; c < dim4; c += 16) {
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                for (uint32_t o = 0; o < cnt; ++o) {
                    const float4 g = grads[(uint64_t)occ[off + o] * dim4 + c];
                    sx += (double)g.x; sy += (double)g.y; sz += (double)g.z; sw += (double)g.w;
                }
                const float4 g = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
                const uint64_t oo = (uint64_t)slot * dim4 + c;
                float4 w = values[oo], x1 = s1[oo], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.kind == MEE_OPT_ADAM) x2 = s2[oo];
                opt_update4(a, w, x1, x2, g);
                values[oo] = w; s1[oo] = x1;
                if (a.kind == MEE_OPT_ADAM) s2[oo] = x2;
            }
        }
        if (inb && tl == 0) { skeys[h] = 0; sval[h] = 0; }
    }
}

// ---- standalone duplicate-key reduction (SPEC.md §4) -------------------------------------------------------
__global__ __launch_bounds__(256) void dedup_fill_kernel(uint32_t n, const uint32_t* __restrict__ hidx,
                                                         const uint32_t* __restrict__ rank,
                                                         const uint32_t* __restrict__ soffs,
                                                         const uint32_t* __restrict__ sgrp, uint32_t* occ,
                                                         int64_t* inverse) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hidx[i];
    if (h != kNoGroup) occ[soffs[h] + rank[i]] = i;
    if (inverse) inverse[i] = h != kNoGroup ? (int64_t)sgrp[h] : -1;
}

__global__ __launch_bounds__(256) void dedup_emit_kernel(uint32_t dim4, const float4* __restrict__ grads,
                                                         const uint32_t* __restrict__ uniq_h, unsigned long long* skeys,
                                                         uint32_t* sval, const uint32_t* __restrict__ soffs,
                                                         const uint32_t* __restrict__ occ, const OpCounters* op,
                                                         int64_t* uniq_out, float4* gsum_out, uint32_t* counts_out) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t nu = op->n_uniq;
    for (uint32_t base = wave * 4; base < nu; base += n_waves * 4) {
        const uint32_t u = base + tile;
        if (u >= nu) continue;
        const uint32_t h = uniq_h[u];
        const uint32_t cnt = sval[h];
        const uint32_t off = soffs[h];
        if (grads && gsum_out) {
            for (uint32_t c = tl; c < dim4; c += 16) {
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                for (uint32_t o = 0; o < cnt; ++o) {
                    const float4 g = grads[(uint64_t)occ[off + o] * dim4 + c];
                    sx += (double)g.x; sy += (double)g.y; sz += (double)g.z; sw += (double)g.w;
                }
                gsum_out[(uint64_t)u * dim4 + c] = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
            }
        }
        if (tl == 0) {
            if (uniq_out) uniq_out[u] = (int64_t)(skeys[h] ^ kBias);
            if (counts_out) counts_out[u] = cnt;
            skeys[h] = 0; sval[h] = 0;
        }
    }
}

// ---- export (SPEC.md §3) -----------------------------------------------------------------------------------
// A wave inspects 64 consecutive slots, compacts the occupied ones with ballot + popcount and reserves its
// output range with one atomic; rows are then copied four at a time (one per tile).
__global__ __launch_bounds__(256) void export_kernel(const int64_t* __restrict__ tkeys, const float4* __restrict__ values,
                                                     const float4* __restrict__ s1, const float4* __restrict__ s2,
                                                     uint64_t capacity, uint32_t dim4, int64_t* keys_out, float4* values_out,
                                                     float4* s1_out, float4* s2_out, uint64_t cap, OpCounters* op) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t s0 = wave * 64; s0 < capacity; s0 += n_waves * 64) {
        const uint64_t s = s0 + lane;
        const int64_t k = s < capacity ? tkeys[s] : kEmpty;
        const uint64_t m = __ballot(!reserved_key(k));
        if (!m) continue;  // wave-uniform
        unsigned long long basev = 0;
        if (lane == 0) basev = atomicAdd(&op->n_export, (unsigned long long)__popcll(m));
        basev = __shfl(basev, 0);
        if (!reserved_key(k)) {
            const uint64_t pos = basev + (uint64_t)__popcll(m & ((1ull << lane) - 1));
            if (keys_out && pos < cap) keys_out[pos] = k;
        }
        uint64_t rest = m;
        uint64_t done = 0;
        while (rest) {  // wave-uniform
            uint64_t mm = rest;
            int p = -1;
            for (int q = 0; q <= tile; ++q) {
                if (mm) { p = __ffsll((unsigned long long)mm) - 1; mm &= mm - 1; } else p = -1;
            }
            const uint64_t pos = basev + done + tile;
            if (p >= 0 && pos < cap) {
                const uint64_t src = (s0 + p) * dim4, dst = pos * dim4;
                for (uint32_t c = tl; c < dim4; c += 16) {
                    if (values_out) values_out[dst + c] = values[src + c];
                    if (s1_out) s1_out[dst + c] = s1[src + c];
                    if (s2_out) s2_out[dst + c] = s2[src + c];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rest &= rest - 1;  // x & (x-1) of 0 stays 0
            done += 4;
        }
    }
}

// =========================================================================================================
// host side
// =========================================================================================================
static hipStream_t as_stream(void* s) { return (hipStream_t)s; }

static int check_batch(const mee_table* t, size_t n, const char* op) {
    if (n > t->max_batch)
        return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: n=%zu exceeds config.max_batch=%llu", op, n, (unsigned long long)t->max_batch);
    return MEE_OK;
}

}  // namespace mee

using namespace mee;

extern "C" {

int mee_abi_version(void) { return MEE_ABI_VERSION; }
const char* mee_last_error(void) { return last_error_buf(); }

int mee_table_destroy(mee_table* t) {
    if (!t) return MEE_OK;
    DeviceGuard g(t->device);
    (void)hipDeviceSynchronize();
    void* dev[] = {t->keys, t->values, t->s1, t->s2, t->skeys, t->sval, t->soffs, t->sgrp, t->sres,
                   t->hidx, t->rank, t->occ, t->uniq_h, t->ctr, t->op};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (t->h_ctr) (void)hipHostFree(t->h_ctr);
    if (t->h_op) (void)hipHostFree(t->h_op);
    delete t;
    return MEE_OK;
}

int mee_table_create(const mee_config* cfg, mee_table** out) {
    if (!cfg || !out) return fail(MEE_ERR_INVALID_ARG, "mee_table_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(mee_config))
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(mee_config));
    if (cfg->capacity == 0 || cfg->dim < 4 || cfg->dim > 1024 || (cfg->dim & 3))
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: capacity must be >0 and dim a multiple of 4 in [4,1024]");
    if (cfg->optimizer > MEE_OPT_ADAM || cfg->initializer > MEE_INIT_UNIFORM)
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: bad optimizer/initializer");
    if (cfg->max_batch == 0 || cfg->max_batch > (1ull << 30))
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: max_batch must be in [1, 2^30]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MEE_ERR_NO_DEVICE, "mee_table_create: no HIP device visible (this backend has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: device %d out of range (have %d)", cfg->device, ndev);
    hipDeviceProp_t prop;
    MEE_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MEE_ERR_NO_DEVICE, "mee_table_create: device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);

    DeviceGuard g(cfg->device);
    if (g.err != hipSuccess) return fail(MEE_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(g.err));

    mee_table* t = new (std::nothrow) mee_table();
    if (!t) return fail(MEE_ERR_OUT_OF_MEMORY, "host allocation failed");
    memset(t, 0, sizeof *t);
    t->device = cfg->device;
    t->nb = (cfg->capacity + kW - 1) / kW;
    t->capacity = t->nb * kW;
    t->dim = cfg->dim; t->dim4 = cfg->dim / 4;
    t->optimizer = cfg->optimizer; t->initializer = cfg->initializer;
    t->max_batch = cfg->max_batch;
    t->default_value = cfg->default_value; t->init_acc = cfg->initial_accumulator;
    t->init_scale = cfg->init_scale; t->init_seed = cfg->init_seed;
    uint64_t S = 1024;
    while (S < 2 * cfg->max_batch) S <<= 1;
    t->S = S; t->smask = S - 1;
    t->find_rounds = 2;
    t->find_grid_cap = 0;
    t->find_nt = 1;

    const uint64_t plane = t->capacity * (uint64_t)t->dim * sizeof(float);
    const uint64_t mb = t->max_batch;
    int rc = MEE_OK;
#define ALLOC(ptr, bytes)                                                                                     \
    do {                                                                                                      \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                                   \
        if (e_ != hipSuccess) {                                                                               \
            rc = fail(MEE_ERR_OUT_OF_MEMORY, "hipMalloc(%llu bytes) for %s: %s", (unsigned long long)(bytes), #ptr, hipGetErrorString(e_)); \
            goto bad;                                                                                         \
        }                                                                                                     \
    } while (0)
    ALLOC(t->keys, t->capacity * sizeof(int64_t));
    ALLOC(t->values, plane);
    t->table_bytes = t->capacity * sizeof(int64_t) + plane;
    if (t->optimizer != MEE_OPT_NONE) { ALLOC(t->s1, plane); t->table_bytes += plane; }
    if (t->optimizer == MEE_OPT_ADAM) { ALLOC(t->s2, plane); t->table_bytes += plane; }
    ALLOC(t->skeys, S * 8); ALLOC(t->sval, S * 4); ALLOC(t->soffs, S * 4); ALLOC(t->sgrp, S * 4); ALLOC(t->sres, S * 8);
    ALLOC(t->hidx, mb * 4); ALLOC(t->rank, mb * 4); ALLOC(t->occ, mb * 4); ALLOC(t->uniq_h, mb * 4);
    ALLOC(t->ctr, sizeof(Counters)); ALLOC(t->op, sizeof(OpCounters));
#undef ALLOC
    t->workspace_bytes = S * 28 + mb * 16 + sizeof(Counters) + sizeof(OpCounters);
    if (hipHostMalloc((void**)&t->h_ctr, sizeof(Counters)) != hipSuccess || hipHostMalloc((void**)&t->h_op, sizeof(OpCounters)) != hipSuccess) {
        rc = fail(MEE_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        goto bad;
    }
    {
        hipError_t e = hipSuccess;
        fill_i64_kernel<<<2048, 256, 0, 0>>>(t->keys, t->capacity, kEmpty);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemsetAsync(t->skeys, 0, S * 8, 0);
        if (e == hipSuccess) e = hipMemsetAsync(t->sval, 0, S * 4, 0);
        if (e == hipSuccess) e = hipMemsetAsync(t->ctr, 0, sizeof(Counters), 0);
        if (e == hipSuccess) e = hipMemsetAsync(t->op, 0, sizeof(OpCounters), 0);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(MEE_ERR_HIP, "table initialisation: %s", hipGetErrorString(e)); goto bad; }
    }
    *out = t;
    return MEE_OK;
bad:
    mee_table_destroy(t);
    return rc;
}

int mee_table_info_get(const mee_table* t, mee_table_info* o) {
    if (!t || !o) return fail(MEE_ERR_INVALID_ARG, "mee_table_info_get: null argument");
    o->capacity = t->capacity; o->n_buckets = t->nb; o->max_batch = t->max_batch;
    o->dim = t->dim; o->optimizer = t->optimizer;
    o->table_bytes = t->table_bytes; o->workspace_bytes = t->workspace_bytes;
    return MEE_OK;
}

int mee_set_tuning(mee_table* t, const char* name, int value) {
    if (!t || !name) return fail(MEE_ERR_INVALID_ARG, "mee_set_tuning: null argument");
    if (!strcmp(name, "find_rounds")) t->find_rounds = value;
    else if (!strcmp(name, "find_grid_cap")) t->find_grid_cap = value;
    else if (!strcmp(name, "find_nt")) t->find_nt = value;
    else return fail(MEE_ERR_INVALID_ARG, "mee_set_tuning: unknown knob '%s'", name);
    return MEE_OK;
}

int mee_clear(mee_table* t, void* stream) {
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_clear: null table");
    DeviceGuard g(t->device);
    fill_i64_kernel<<<2048, 256, 0, as_stream(stream)>>>(t->keys, t->capacity, kEmpty);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemsetAsync(&t->ctr->size, 0, sizeof(unsigned long long), as_stream(stream)));
    return MEE_OK;
}

int mee_find(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find: null argument");
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    int R = t->find_rounds;
    if (t->dim4 != 16 && R > 4) R = 4;
    if (t->dim4 != 16 && t->dim4 != 32 && R > 2) R = 2;
    R = R >= 8 ? 8 : R >= 4 ? 4 : R >= 2 ? 2 : 1;
    const unsigned grid = grid_for(n, 4u * 4u * (unsigned)R, t->find_grid_cap > 0 ? (unsigned)t->find_grid_cap : (1u << 22));
#define FIND1(D4, RR, NT) find_kernel<D4, RR, NT><<<grid, 256, 0, st>>>(t->keys, (const f32x4*)t->values, t->nb, d_keys, n, (f32x4*)d_out, d_found, t->default_value, t->dim4)
#define FIND(D4, RR) do { switch (t->find_nt & 3) { case 0: FIND1(D4, RR, 0); break; case 1: FIND1(D4, RR, 1); break; case 2: FIND1(D4, RR, 2); break; default: FIND1(D4, RR, 3); } } while (0)
    if (t->dim4 == 16) { if (R == 8) FIND(16, 8); else if (R == 4) FIND(16, 4); else if (R == 2) FIND(16, 2); else FIND(16, 1); }
    else if (t->dim4 == 32) { if (R == 4) FIND(32, 4); else if (R == 2) FIND(32, 2); else FIND(32, 1); }
    else { if (R == 2) FIND(0, 2); else FIND(0, 1); }
#undef FIND
#undef FIND1
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

static int upsert_common(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found,
                         void* stream, bool claim, const char* name) {
    if (!t || (n && (!d_keys || !d_values))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (int rc = check_batch(t, n, name)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    const unsigned gl = grid_for(n, 256, 1u << 22), gt = grid_for(n, 16, 1u << 16);
    group_kernel<false><<<gl, 256, 0, st>>>(d_keys, nn, t->skeys, t->smask, t->sval, t->sgrp, t->hidx, t->rank, t->uniq_h, t->ctr, t->op);
    if (claim)
        upsert_kernel<true><<<gt, 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4, d_keys,
                                                (const float4*)d_values, nn, t->hidx, t->sval, nullptr, t->optimizer, t->init_acc, t->ctr);
    else
        upsert_kernel<false><<<gt, 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4, d_keys,
                                                 (const float4*)d_values, nn, t->hidx, t->sval, d_found, t->optimizer, t->init_acc, t->ctr);
    group_reset_kernel<<<gl, 256, 0, st>>>(t->hidx, nn, t->skeys, t->sval);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_insert(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, void* stream) {
    return upsert_common(t, d_keys, d_values, n, nullptr, stream, true, "mee_insert");
}
int mee_assign(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream) {
    return upsert_common(t, d_keys, d_values, n, d_found, stream, false, "mee_assign");
}

int mee_find_or_insert(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_or_insert: null argument");
    if (int rc = check_batch(t, n, "mee_find_or_insert")) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    const unsigned gl = grid_for(n, 256, 1u << 22), gt = grid_for(n, 16, 1u << 16);
    group_kernel<false><<<gl, 256, 0, st>>>(d_keys, nn, t->skeys, t->smask, t->sval, t->sgrp, t->hidx, t->rank, t->uniq_h, t->ctr, t->op);
    ensure_kernel<<<gt, 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4, d_keys, nn, t->hidx,
                                      t->sval, t->sres, t->optimizer, t->init_acc, t->initializer, t->init_scale, t->init_seed,
                                      t->default_value, t->ctr);
    gather_group_kernel<<<gt, 256, 0, st>>>((const float4*)t->values, t->dim4, nn, t->hidx, t->sres, (float4*)d_out, d_found, t->default_value);
    group_reset_kernel<<<gl, 256, 0, st>>>(t->hidx, nn, t->skeys, t->sval);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_export(const mee_table* t, int64_t* d_keys_out, float* d_values_out, float* d_state1_out, float* d_state2_out,
               size_t cap, size_t* n_out, void* stream) {
    if (!t || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_export: null argument");
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    MEE_HIP(hipMemsetAsync(&t->op->n_export, 0, sizeof(unsigned long long), st));
    export_kernel<<<grid_for(t->capacity, 256, 256 * 16), 256, 0, st>>>(
        t->keys, (const float4*)t->values, (const float4*)t->s1, (const float4*)t->s2, t->capacity, t->dim4, d_keys_out,
        (float4*)d_values_out, t->s1 ? (float4*)d_state1_out : nullptr, t->s2 ? (float4*)d_state2_out : nullptr, cap, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)t->h_op->n_export;
    return MEE_OK;
}

static int read_counters(const mee_table* t, void* stream) {
    DeviceGuard g(t->device);
    MEE_HIP(hipMemcpyAsync(t->h_ctr, t->ctr, sizeof(Counters), hipMemcpyDeviceToHost, as_stream(stream)));
    MEE_HIP(hipStreamSynchronize(as_stream(stream)));
    return MEE_OK;
}
int mee_size(const mee_table* t, size_t* n_out, void* stream) {
    if (!t || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_size: null argument");
    if (int rc = read_counters(t, stream)) return rc;
    *n_out = (size_t)t->h_ctr->size;
    return MEE_OK;
}
int mee_status(const mee_table* t, uint32_t* bits_out, void* stream) {
    if (!t || !bits_out) return fail(MEE_ERR_INVALID_ARG, "mee_status: null argument");
    if (int rc = read_counters(t, stream)) return rc;
    *bits_out = t->h_ctr->status;
    return MEE_OK;
}
int mee_clear_status(mee_table* t, void* stream) {
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_clear_status: null table");
    DeviceGuard g(t->device);
    MEE_HIP(hipMemsetAsync(&t->ctr->status, 0, sizeof(uint32_t), as_stream(stream)));
    return MEE_OK;
}

static int apply_common(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, const OptArgs& a, void* stream,
                        const char* name) {
    if (!t || (n && (!d_keys || !d_grads))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (t->optimizer != a.kind) return fail(MEE_ERR_UNSUPPORTED, "%s: table was created with optimizer=%u", name, t->optimizer);
    if (int rc = check_batch(t, n, name)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    const unsigned gl = grid_for(n, 256, 1u << 22), gt = grid_for(n, 16, 1u << 16);
    MEE_HIP(hipMemsetAsync(t->op, 0, 8, st));  // n_uniq, n_occ
    group_kernel<true><<<gl, 256, 0, st>>>(d_keys, nn, t->skeys, t->smask, t->sval, t->sgrp, t->hidx, t->rank, t->uniq_h, t->ctr, t->op);
    group_offsets_kernel<false><<<gl, 256, 0, st>>>(t->uniq_h, t->sval, t->soffs, t->op);
    apply_single_kernel<<<gt, 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4, d_keys,
                                            (const float4*)d_grads, nn, t->hidx, t->rank, t->sval, t->soffs, t->occ, a);
    apply_multi_kernel<<<gt, 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4,
                                           (const float4*)d_grads, t->uniq_h, t->skeys, t->sval, t->soffs, t->occ, t->op, a);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_apply_adagrad(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps, void* stream) {
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adagrad");
}
int mee_apply_adam(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1, float beta2,
                   float eps, uint64_t step, void* stream) {
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adam: step must be >= 1");
    OptArgs a{};
    a.kind = MEE_OPT_ADAM; a.eps = eps;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr * sqrt(bc2) / bc1);  // SPEC.md §4
    a.omb1 = 1.0f - beta1; a.omb2 = 1.0f - beta2;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adam");
}

int mee_dedup_sum(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, int64_t* d_uniq_out, float* d_gsum_out,
                  uint32_t* d_counts_out, int64_t* d_inverse_out, size_t* n_unique_out, void* stream) {
    if (!t || !n_unique_out || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_dedup_sum: null argument");
    if (int rc = check_batch(t, n, "mee_dedup_sum")) return rc;
    *n_unique_out = 0;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    const unsigned gl = grid_for(n, 256, 1u << 22), gt = grid_for(n, 16, 1u << 16);
    MEE_HIP(hipMemsetAsync(t->op, 0, 8, st));
    group_kernel<true><<<gl, 256, 0, st>>>(d_keys, nn, t->skeys, t->smask, t->sval, t->sgrp, t->hidx, t->rank, t->uniq_h, t->ctr, t->op);
    group_offsets_kernel<true><<<gl, 256, 0, st>>>(t->uniq_h, t->sval, t->soffs, t->op);
    dedup_fill_kernel<<<gl, 256, 0, st>>>(nn, t->hidx, t->rank, t->soffs, t->sgrp, t->occ, d_inverse_out);
    dedup_emit_kernel<<<gt, 256, 0, st>>>(t->dim4, (const float4*)d_grads, t->uniq_h, t->skeys, t->sval, t->soffs, t->occ, t->op,
                                          d_uniq_out, (float4*)d_gsum_out, d_counts_out);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_unique_out = t->h_op->n_uniq;
    return MEE_OK;
}

}  // extern "C"

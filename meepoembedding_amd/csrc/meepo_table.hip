// meepo_table.hip — HBM-resident open-addressing table for gfx950: life cycle (create / destroy / tuning / clear), the mutators (insert,
// assign, remove, find_or_insert and their kernels), the entry points of the sparse optimizers (the kernels: meepo_apply.hip) and the
// standalone duplicate reduction — the table half of the C-ABI.  The lookups: meepo_find.hip; export, size, rehash, statistics:
// meepo_export.hip; dedup_keys and assign's elections: meepo_dedup.hip.
//
// Reference anchor: /root/reference/README.md:2 ("dynamic lookuptable-style Embedding … Supports GPU …"); the
// snapshot has no code, so semantics come from SPEC.md (§2 table, §3 operators, §4 optimizers).
//
// HBM layout (all slot-indexed, one allocation each):
//   keys   int64[capacity]          16 keys = one 128-B line = one bucket
//   values fp32 [capacity][dim]     row of slot s at values + s*dim  (256 B for dim 64)
//   s1,s2  fp32 [capacity][dim]     optimizer planes (acc | m, v), only if configured
// Per-batch scratch ("group table"): an open-addressing set of the batch's distinct keys sized ≥ 2·max_batch (stays in L2 / Infinity
// Cache): the last-wins election of insert among positions that find their key present.  (The optimizers, dedup_keys, assign and — round 5 —
// dedup_sum left it for the partition + block-local LDS tables of meepo_apply.hip / meepo_dedup.hip.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "meepo_apply_part.h"

namespace mee {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

const Roctx& roctx_api() {
    static const Roctx api = [] {
        Roctx r;
        const char* env = getenv("MEE_ROCTX");
        if (env && env[0] == '0') return r;
        void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return r;
        *(void**)(&r.push) = dlsym(h, "roctxRangePushA");
        *(void**)(&r.pop) = dlsym(h, "roctxRangePop");
        if (!r.push || !r.pop) r.push = nullptr, r.pop = nullptr;
        return r;
    }();
    return api;
}

}  // namespace mee

namespace mee {

// =========================================================================================================
// kernels
// =========================================================================================================

// Zeroing of the small device-side counter blocks.  Not hipMemsetAsync: captured in a hipGraph, the memset node did not
// reliably zero the block on later replays once other copies had run in between (counters kept their old values, the apply
// lists then overflowed their buffers) — a kernel node of our own has no such state.
__global__ void zero_words_kernel(uint32_t* p, uint32_t n_words) {
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) p[i] = 0u;
}
void zero_words(void* p, size_t bytes, hipStream_t st) { zero_words_kernel<<<1, 64, 0, st>>>((uint32_t*)p, (uint32_t)(bytes / 4)); }

__global__ void fill_i64_kernel(int64_t* p, uint64_t n, int64_t v) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

void fill_keys(int64_t* p, uint64_t n, int64_t v, hipStream_t st) { fill_i64_kernel<<<2048, 256, 0, st>>>(p, n, v); }

// ---- group table: one entry per distinct key of the batch ---------------------------------------------------
__device__ __forceinline__ uint32_t group_claim(const GroupTable& g, int64_t key, bool& claimed) {
    const unsigned long long bk = (unsigned long long)key ^ kBias;  // != 0 because key != kEmpty
    uint32_t h = (uint32_t)(mix64b((uint64_t)key) & g.smask);
    while (true) {  // one CAS per probe step: callers are already one lane per (block, key), so no pre-read is needed
        const unsigned long long cur = atomicCAS(&g.ent[2 * (uint64_t)h], 0ull, bk);
        claimed = cur == 0;
        if (cur == 0 || cur == bk) return h;
        h = (h + 1) & (uint32_t)g.smask;
    }
}

constexpr uint32_t kEpochWrap = (1u << 31) - 16;   // batch numbers (mee_table::epoch) start over here

// One lane per batch position.  Occurrences of the same key inside a 256-thread block are first combined in an
// LDS hash table keyed by the key itself; ONE lane per (block, key) then talks to the global group table.  A hot key
// that is 8 % of the batch costs ~n/256 global accesses instead of 0.08 n serialised on one L2 line, and the block
// whose CAS claimed the entry needs no counting atomic at all.
// (The table holds max(lo, hi) = 1 + the highest batch position of the key: the last occurrence wins.)
__global__ __launch_bounds__(256) void group_last_kernel(const int64_t* __restrict__ keys, uint32_t n, GroupTable g, BatchScratch bs,
                                                         Counters* ctr, const uint8_t* __restrict__ skip, uint32_t epoch, const uint32_t* __restrict__ gate) {
    if (gate && *gate != epoch) return;   // insert: no position of this batch found its key present -> no election (grid-uniform)
    constexpr int BLOCK = 256;
    constexpr int kLds = 2 * BLOCK;   // block-local aggregation table (BLOCK threads -> at most BLOCK distinct keys: half full at worst)
    constexpr int kLdsShift = 55;
    __shared__ unsigned long long lkey[kLds];
    __shared__ uint32_t lval[kLds], lh[kLds];
    for (int j = threadIdx.x; j < kLds; j += BLOCK) { lkey[j] = 0; lval[j] = 0; }
    __syncthreads();
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool inb = i < n;
    const int64_t key = inb ? keys[i] : 0;
    const bool skipped = inb && skip && skip[i];  // position already served by an earlier pass (find_or_insert)
    const bool valid = inb && !skipped && !reserved_key(key);
    uint32_t slot = 0;
    bool inserter = false;
    if (valid) {
        const unsigned long long bk = (unsigned long long)key ^ kBias;
        slot = (uint32_t)(mix64((uint64_t)key) >> kLdsShift);  // log2(kLds) bits
        while (true) {
            const unsigned long long old = atomicCAS(&lkey[slot], 0ull, bk);
            if (old == 0) { inserter = true; break; }
            if (old == bk) break;
            slot = (slot + 1) & (kLds - 1);
        }
        atomicMax(&lval[slot], i + 1);
    }
    __syncthreads();
    if (inserter) {
        bool claimed;
        const uint32_t h = group_claim(g, key, claimed);
        lh[slot] = h;
        // the block whose CAS created the entry parks its candidate in the hi half with a plain store; only later arrivals
        // pay an atomic.  Readers take max(lo, hi).  (Unique keys: one atomic per key instead of two.)
        if (claimed) sv_half(g, h)[1] = lval[slot];
        else atomicMax(&sv_half(g, h)[0], lval[slot]);
    }
    __syncthreads();
    if (inb) {
        bs.hidx[i] = valid ? lh[slot] : kNoGroup;
        if (!valid && !skipped && key == kReclaimed) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_RESERVED_KEY);  // EMPTY = padding, silent
    }
}

// ---- insert without an election pass for batches that need none (SPEC.md §3 insert; last occurrence of a key wins) -----------------
// insert_direct_kernel, every position: probe with claim.  A position whose CAS CREATES its key is alone with it so far and writes its
// row (+ initial optimizer state) at once: a batch of distinct new keys — populating a table, a growing vocabulary — is finished after
// this one kernel with ONE atomic per key.  A position that finds its key present (stored before the batch, or created by another
// occurrence in it) writes nothing, keeps its slot and raises the batch's election flag (ctr->election = epoch, a plain store).
// Only then do the three gated kernels do anything: group_last_kernel over the "present" positions, insert_join_kernel (every
// creator looks its key up in the group table — read-only — and joins the election if other occurrences registered there), and
// insert_settle_kernel (the highest position of each registered key rewrites the row at the slot it kept; rows are plain overwrites
// and the kernel boundary orders them behind the creators' optimistic writes).  `made[i]` = 1 for creators (the group pass skips them).
template <int DIM4>
__global__ __launch_bounds__(256) void insert_direct_kernel(int64_t* tkeys, float4* values, float4* s1, float4* s2, uint64_t nb, uint32_t dim4_rt,
                                                            const int64_t* __restrict__ keys, const float4* __restrict__ vals, uint32_t n,
                                                            const uint8_t* __restrict__ skip, uint8_t* __restrict__ made, long long* __restrict__ slotof,
                                                            uint32_t* __restrict__ hidx, uint32_t optimizer, float init_acc, Counters* ctr,
                                                            uint32_t* hits, uint32_t epoch) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    constexpr int C = DIM4 ? DIM4 / 16 : 1;
    constexpr int R = 2;   // positions in flight per tile: keys, rows and first bucket lines of both are requested up front
    for (uint32_t base = wave * 4 * R; base < n; base += n_waves * 4 * R) {
        int64_t key[R], slot[R];
        bool act[R], is_new[R];
        float4 row[R][C];
        const int64_t kmine = (lane < 4 * R && base + lane < n) ? keys[base + lane] : kEmpty;   // one coalesced load for the wave step
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + r * 4 + tile;
            key[r] = __shfl(kmine, r * 4 + tile);
            act[r] = i < n && !reserved_key(key[r]) && !(skip && skip[i]);
            if (i < n && tl == 0 && key[r] == kReclaimed && !(skip && skip[i])) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_RESERVED_KEY);
            if constexpr (DIM4 != 0) {
                if (act[r]) {   // the row is read exactly once: stream it, and request it before the probe
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vals) + (uint64_t)i * DIM4 + c * 16 + tl);
                        row[r][c] = make_float4(v.x, v.y, v.z, v.w);
                    }
                }
            }
        }
        bool any_full = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bool full;
            slot[r] = tile_locate<true, true>(tkeys, nb, key[r], act[r], tile, tl, is_new[r], full);
            any_full = any_full || full;
        }
        bool present = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = base + r * 4 + tile;
            if (i >= n) continue;
            const bool created = act[r] && slot[r] >= 0 && is_new[r];
            if (created) {
                const float4 ia = make_float4(init_acc, init_acc, init_acc, init_acc), z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (DIM4 != 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const uint64_t o = (uint64_t)slot[r] * DIM4 + c * 16 + tl;
                        values[o] = row[r][c];
                        if (optimizer == MEE_OPT_ADAGRAD) s1[o] = ia;
                        if (optimizer == MEE_OPT_ADAM) { s1[o] = z4; s2[o] = z4; }
                    }
                } else {
                    for (uint32_t c = tl; c < dim4; c += 16) {
                        const uint64_t o = (uint64_t)slot[r] * dim4 + c;
                        values[o] = vals[(uint64_t)i * dim4 + c];
                        if (optimizer == MEE_OPT_ADAGRAD) s1[o] = ia;
                        if (optimizer == MEE_OPT_ADAM) { s1[o] = z4; s2[o] = z4; }
                    }
                }
            }
            if (tl == 0) {
                if (created && hits) hits[slot[r]] = 0;
                made[i] = created || !act[r] || slot[r] < 0;   // 1 = takes no part in the group pass (creator, skipped, reserved key, table full)
                slotof[i] = slot[r];
                hidx[i] = kNoGroup;                              // the group pass fills this in for the positions it takes
            }
            present = present || (act[r] && slot[r] >= 0 && !is_new[r]);
        }
        if (__any(present) && lane == 0) ctr->election = epoch;            // plain store: every writer stores the same value
        if (__any(any_full) && lane == 0) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_TABLE_FULL);
    }
}

// creators join the election of their key if other occurrences registered it (read-only lookup of the group table built by the gated
// group pass: a creator whose key nobody else holds finds an empty entry at once)
__global__ __launch_bounds__(256) void insert_join_kernel(const int64_t* __restrict__ keys, uint32_t n, const uint8_t* __restrict__ made,
                                                          const long long* __restrict__ slotof, GroupTable g, uint32_t* __restrict__ hidx,
                                                          const Counters* ctr, uint32_t epoch) {
    if (ctr->election != epoch) return;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!made[i] || slotof[i] < 0) continue;
        const int64_t key = keys[i];
        if (reserved_key(key)) continue;
        const unsigned long long bk = (unsigned long long)key ^ kBias;
        uint32_t h = (uint32_t)(mix64b((uint64_t)key) & g.smask);
        for (uint64_t step = 0; step <= g.smask; ++step) {
            const unsigned long long cur = g.ent[2 * (uint64_t)h];
            if (cur == 0) break;        // nobody else holds this key
            if (cur == bk) { atomicMax(&sv_half(g, h)[0], i + 1); hidx[i] = h; break; }
            h = (h + 1) & (uint32_t)g.smask;
        }
    }
}

// the highest registered position of each key rewrites the row at the slot it kept from insert_direct_kernel, then releases the entry
__global__ __launch_bounds__(256) void insert_settle_kernel(float4* values, uint32_t dim4, const float4* __restrict__ vals, uint32_t n,
                                                            const long long* __restrict__ slotof, const uint32_t* __restrict__ hidx, GroupTable g,
                                                            const Counters* ctr, uint32_t epoch) {
    if (ctr->election != epoch) return;
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        const uint32_t h = i < n ? hidx[i] : kNoGroup;
        if (h == kNoGroup) continue;
        uint32_t lo, hi;
        sv_load(g, h, lo, hi);   // one 8-byte load: the complete pair, or zeros once the winner released it
        if (max(lo, hi) != i + 1) continue;
        const long long slot = slotof[i];
        if (slot >= 0)
            for (uint32_t c = tl; c < dim4; c += 16) values[(uint64_t)slot * dim4 + c] = vals[(uint64_t)i * dim4 + c];
        if (tl == 0) group_release_entry(g, h);
    }
}

// ---- admission policy (SPEC.md §3 find_or_insert_admit): a count-min sketch of how often ABSENT keys were asked for ------------
__device__ __forceinline__ uint32_t sketch_index(int64_t key, int row, uint32_t log2w) {
    constexpr uint64_t A[3] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
    return (uint32_t)(mix64((uint64_t)key ^ A[row]) >> (64 - log2w)) + ((uint32_t)row << log2w);
}
// every position whose key is absent (found byte 0, key not reserved) adds 1 to its three counters
__global__ __launch_bounds__(256) void sketch_add_kernel(const int64_t* __restrict__ keys, const uint8_t* __restrict__ found, uint64_t n,
                                                         uint32_t* sketch, uint32_t log2w) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t k = keys[i];
        if (found[i] || reserved_key(k)) continue;
#pragma unroll
        for (int r = 0; r < 3; ++r) atomicAdd(&sketch[sketch_index(k, r, log2w)], 1u);
    }
}
// after all additions of the batch: skip[i] = 1 unless the position's key is absent AND its estimate reached min_count
__global__ __launch_bounds__(256) void sketch_decide_kernel(const int64_t* __restrict__ keys, const uint8_t* __restrict__ found, uint64_t n,
                                                            const uint32_t* __restrict__ sketch, uint32_t log2w, uint32_t min_count,
                                                            uint8_t* __restrict__ skip) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t k = keys[i];
        uint8_t s = 1;
        if (!found[i] && !reserved_key(k)) {
            uint32_t est = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 3; ++r) est = min(est, sketch[sketch_index(k, r, log2w)]);
            s = est >= min_count ? 0 : 1;
        }
        skip[i] = s;
    }
}
__global__ void sketch_decay_kernel(uint32_t* sketch, uint64_t n, uint32_t shift) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        sketch[i] = shift >= 32 ? 0u : sketch[i] >> shift;
}

// ---- remove (SPEC.md §3) -----------------------------------------------------------------------------------
// Two kernels so that every occurrence of a duplicate key reports the state before the call: locate (read-only,
// found + slot per position), then tombstone (idempotent stores of RECLAIMED).
// probe only (remove's first half, mee_locate): the find kernel's shape without the rows — 4R positions per wave step, keys by one
// coalesced load, the R first bucket lines requested together, the slots stored by one coalesced store per wave step.  R = 4:
// 1M keys of a 100M-key table in 56 / 44 / 50 us at R = 2 / 4 / 8 (a probe has few bytes per key: it needs more keys in flight than find)
template <int R>
__global__ __launch_bounds__(256) void remove_locate_kernel(const int64_t* __restrict__ tkeys, uint64_t nb,
                                                            const int64_t* __restrict__ keys, uint32_t n, long long* slot_out,
                                                            uint8_t* found, Counters* ctr) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    constexpr int KPW = 4 * R;
    for (uint32_t base = wave * KPW; base < n; base += n_waves * KPW) {
        int64_t key[R], slot[R], kb[R];
        uint64_t b[R];
        bool valid[R];
        const bool mine = lane < KPW && base + lane < n;
        const int64_t kmine = mine ? keys[base + lane] : kEmpty;
        if (mine && kmine == kReclaimed) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_RESERVED_KEY);   // EMPTY = padding, silent
#pragma unroll
        for (int r = 0; r < R; ++r) {
            key[r] = __shfl(kmine, r * 4 + tile);
            valid[r] = base + r * 4 + tile < n && !reserved_key(key[r]);
            b[r] = bucket_of(key[r], nb);
            kb[r] = valid[r] ? tkeys[b[r] * kW + tl] : kEmpty;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            slot[r] = -1;
            bool pend = valid[r];
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
        long long my_slot = -1;   // lane j < 4R collects the slot of position base + j (round j / 4, tile j % 4)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long v = __shfl((long long)slot[r], (lane & 3) * kW);
            if ((lane >> 2) == r) my_slot = v;
        }
        if (mine) {
            slot_out[base + lane] = my_slot;
            if (found) found[base + lane] = my_slot >= 0;
        }
    }
}
__global__ __launch_bounds__(256) void remove_mark_kernel(int64_t* tkeys, const long long* __restrict__ slot_in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && slot_in[i] >= 0) tkeys[slot_in[i]] = kReclaimed;
}

// ---- find_or_insert (SPEC.md §3), middle pass: every position the find pass left missing claims its key's slot (or finds
// it, when another occurrence of the key got there first: the CAS decides the one creator); the creating tile writes the
// hashed initial row, the initial optimizer state and a zero hit counter.  A wave with nothing missing leaves at once —
// the steady state of a trained vocabulary costs one pass over keys + found.  The rows are read by a last find pass.
template <int P>   // positions per wave step (64; 32 and 16 measured no faster on batches of new keys — the pass is bound by its claims — and
                   // 3-7 % slower when nearly every key is present: tools/foi_bench.py)
__global__ __launch_bounds__(256) void ensure_direct_kernel(int64_t* tkeys, float4* values, float4* s1, float4* s2, uint64_t nb,
                                                            uint32_t dim4, const int64_t* __restrict__ keys, uint32_t n,
                                                            const uint8_t* __restrict__ found, uint32_t optimizer, float init_acc,
                                                            uint32_t initializer, float init_scale, uint64_t init_seed,
                                                            float default_value, Counters* ctr, uint32_t* hits, float4* __restrict__ out,
                                                            long long* __restrict__ slots_out = nullptr, long long handle_tag = 0) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t base = wave * P; base < n; base += n_waves * P) {
        // P positions per wave step: one coalesced load of keys and found bytes, then the missing ones four at a time
        const uint32_t i = base + lane;
        const bool mine = lane < P && i < n;
        const int64_t k = mine ? keys[i] : kEmpty;
        const bool miss = mine && found[i] == 0;
        if (miss && k == kReclaimed) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_RESERVED_KEY);  // EMPTY = padding, silent
        uint64_t rest = __ballot(miss && !reserved_key(k));
        while (rest) {  // wave-uniform
            uint64_t mm = rest;
            int p = -1;
            for (int q = 0; q <= tile; ++q) {
                if (mm) { p = __ffsll((unsigned long long)mm) - 1; mm &= mm - 1; } else p = -1;
            }
            const int64_t key = __shfl(k, p >= 0 ? p : 0);
            bool is_new, full;
            const int64_t slot = tile_locate<true, true>(tkeys, nb, key, p >= 0, tile, tl, is_new, full);
            if (p >= 0 && slot >= 0) {
                // A position that was missing and whose key is stored now — created by this tile or by another occurrence in this batch —
                // returns the key's initial row, which is a function of the key alone: every such tile writes it straight into `out`
                // (bit-identical to what the creator stores), so no third pass has to read the created rows back.
                for (uint32_t c = tl; c < dim4; c += 16) {
                    const float4 row = initial_row4(key, c * 4, initializer, init_scale, init_seed, default_value);
                    if (out) out[(uint64_t)(base + p) * dim4 + c] = row;
                    if (is_new) {
                        values[(uint64_t)slot * dim4 + c] = row;
                        if (optimizer == MEE_OPT_ADAGRAD) s1[(uint64_t)slot * dim4 + c] = make_float4(init_acc, init_acc, init_acc, init_acc);
                        if (optimizer == MEE_OPT_ADAM) {
                            s1[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                            s2[(uint64_t)slot * dim4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                if (is_new && hits && tl == 0) hits[slot] = 0;
                if (slots_out && tl == 0) slots_out[base + p] = slot | handle_tag;   // the located variant: the find pass left -1 here
            }
            const uint64_t fm = __ballot(full);
            if (lane == 0 && fm) atomicOr(&ctr->status, (uint32_t)MEE_STATUS_TABLE_FULL);
#pragma unroll
            for (int q = 0; q < 4; ++q) rest &= rest - 1;
        }
    }
}

// =========================================================================================================
// host side
// =========================================================================================================
TableView table_view(const mee_table* t) {
    return TableView{t->device, t->keys, t->values, t->nb, t->dim, t->dim4, t->default_value, t->generation, t->s1, t->s2, t->optimizer,
                     t->initializer, t->init_scale, t->init_acc, t->init_seed, &t->ctr->status, t->hits};
}

// SPEC.md §2: the bucket count is prime, so every double-hashing stride visits all buckets
uint64_t next_prime(uint64_t n) {
    if (n <= 2) return 2;
    if (!(n & 1)) ++n;
    for (;; n += 2) {
        bool prime = true;
        for (uint64_t d = 3; d * d <= n; d += 2)
            if (n % d == 0) { prime = false; break; }
        if (prime) return n;
    }
}

// the next batch number (insert's election flag carries it: a flag left by an earlier batch reads as "not raised")
static void next_epoch(mee_table* t, hipStream_t) {
    if (++t->epoch >= kEpochWrap) t->epoch = 1;
}
}  // namespace mee
extern "C" int mee_apply_discard(mee_table* t, void* stream);
namespace mee {
// A mutator (or mee_reserve) that needs the per-batch scratch while the partition a TRAINING FORWARD left (mee_find*_located_prepare) is still
// pending — its backward has not come yet; an eviction, a growth step or a second lookup in between — drops that partition itself: the apply
// that follows finds nothing prepared and partitions its batch again (same results, one partition launch more).  `stream`: where the mutator
// runs.  A partition the caller asked for explicitly (mee_apply_prepare, perhaps on another stream) is the caller's to finish or discard.
int check_batch(mee_table* t, size_t n, const char* op, void* stream, bool needs_group_table, bool drop_pending) {
    if (t->prepared_n && needs_group_table) {
        if (!drop_pending || !t->prepared_by_forward)
            return fail(MEE_ERR_INVALID_ARG, "%s: a prepared apply is pending on this table (finish it with mee_apply_* or mee_apply_discard)", op);
        if (int rc = mee_apply_discard(t, stream)) return rc;
    }
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
    MEE_RANGE("mee_table_destroy");
    if (!t) return MEE_OK;
    DeviceGuard g(t->device);
    (void)hipDeviceSynchronize();
    float* planes[] = {t->values, t->s1, t->s2};
    for (float* p : planes)
        if (p) { if (t->value_memory == MEE_MEM_HOST_PINNED) (void)hipHostFree(p); else (void)hipFree(p); }
    void* dev[] = {t->keys, t->hits, t->sketch, t->g.ent, t->g.sres, t->bs.hidx, t->bs.occ, t->bs.fmask, t->bs.gacc, t->ctr, t->op};
    for (void* p : dev) if (p) (void)hipFree(p);
    bucket_scratch_free(t);
    if (t->h_ctr) (void)hipHostFree(t->h_ctr);
    if (t->h_op) (void)hipHostFree(t->h_op);
    delete t;
    return MEE_OK;
}

int mee_table_create(const mee_config* cfg, mee_table** out) {
    MEE_RANGE("mee_table_create");
    if (!cfg || !out) return fail(MEE_ERR_INVALID_ARG, "mee_table_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(mee_config))
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(mee_config));
    if (cfg->capacity == 0 || cfg->dim < 4 || cfg->dim > 1024 || (cfg->dim & 3))
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: capacity must be >0 and dim a multiple of 4 in [4,1024]");
    if (cfg->optimizer > MEE_OPT_ADAM || cfg->initializer > MEE_INIT_UNIFORM || cfg->value_memory > MEE_MEM_HOST_PINNED ||
        (cfg->flags & ~(uint32_t)(MEE_FLAG_TRACK_HITS | MEE_FLAG_ADMISSION)) != 0)
        return fail(MEE_ERR_INVALID_ARG, "mee_table_create: bad optimizer/initializer/value_memory");
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
    t->nb = next_prime((cfg->capacity + kW - 1) / kW);
    t->capacity = t->nb * kW;
    t->dim = cfg->dim; t->dim4 = cfg->dim / 4;
    t->optimizer = cfg->optimizer; t->initializer = cfg->initializer; t->value_memory = cfg->value_memory;
    t->max_batch = cfg->max_batch;
    t->default_value = cfg->default_value; t->init_acc = cfg->initial_accumulator;
    t->init_scale = cfg->init_scale; t->init_seed = cfg->init_seed;
    uint64_t S = 1024;
    while (S < 2 * cfg->max_batch) S <<= 1;
    t->S = S; t->g.smask = S - 1;
    t->find_rounds = 0;  // auto: 2 for dim 64, 1 for wider rows
    t->find_grid_cap = 0;
    t->find_nt = -1;  // auto

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
    // value/state planes: HBM, or pinned device-mapped host DRAM for a cold tier (same kernels, rows travel over PCIe)
#define ALLOC_PLANE(ptr)                                                                                      \
    do {                                                                                                      \
        if (t->value_memory == MEE_MEM_HOST_PINNED) {                                                         \
            hipError_t e_ = hipHostMalloc((void**)&(ptr), plane, hipHostMallocMapped | hipHostMallocPortable); \
            if (e_ != hipSuccess) {                                                                           \
                rc = fail(MEE_ERR_OUT_OF_MEMORY, "hipHostMalloc(%llu bytes) for %s: %s", (unsigned long long)plane, #ptr, hipGetErrorString(e_)); \
                goto bad;                                                                                     \
            }                                                                                                 \
        } else ALLOC(ptr, plane);                                                                             \
        t->table_bytes += plane;                                                                              \
    } while (0)
    ALLOC(t->keys, t->capacity * sizeof(int64_t));
    t->table_bytes = t->capacity * sizeof(int64_t);
    if (cfg->flags & MEE_FLAG_TRACK_HITS) { ALLOC(t->hits, t->capacity * sizeof(uint32_t)); t->table_bytes += t->capacity * sizeof(uint32_t); }
    if (cfg->flags & MEE_FLAG_ADMISSION) {   // SPEC.md §3: W = max(2^12, 2^ceil(log2(capacity / 16)))
        t->sketch_log2w = 12;
        while ((1ull << t->sketch_log2w) < (t->capacity + 15) / 16 && t->sketch_log2w < 30) ++t->sketch_log2w;
        ALLOC(t->sketch, 3ull * sizeof(uint32_t) << t->sketch_log2w);
        t->table_bytes += 3ull * sizeof(uint32_t) << t->sketch_log2w;
    }
    ALLOC_PLANE(t->values);
    if (t->optimizer != MEE_OPT_NONE) ALLOC_PLANE(t->s1);
    if (t->optimizer == MEE_OPT_ADAM) ALLOC_PLANE(t->s2);
#undef ALLOC_PLANE
    ALLOC(t->g.ent, S * 16); ALLOC(t->g.sres, S * 8);
    ALLOC(t->bs.hidx, mb * 4); ALLOC(t->bs.occ, mb * 4); ALLOC(t->bs.fmask, mb);
    // fp64 partial rows of the bucketed apply's long runs (meepo_apply.hip): it cuts runs into chunks of 8 sources and quads of four chunks — one row per quad of a
    // run longer than 32: at most m / 32 per slab of m sources, slabs and merge passes of split buckets together (+ 32 rows per single-key merge pass); sized with room
    t->max_part = t->optimizer != MEE_OPT_NONE ? mb / 4 + mb / 32 + 4096 : 0;
    t->bs.max_part = (uint32_t)t->max_part;
    if (t->optimizer != MEE_OPT_NONE) ALLOC(t->bs.gacc, t->max_part * (uint64_t)t->dim * sizeof(double));
    ALLOC(t->ctr, sizeof(Counters)); ALLOC(t->op, sizeof(OpCounters));
#undef ALLOC
    t->workspace_bytes = S * 24 + mb * 9 + (t->bs.gacc ? t->max_part * (uint64_t)t->dim * sizeof(double) : 0) + sizeof(Counters) + sizeof(OpCounters);
    if ((rc = bucket_scratch_alloc(t)) != MEE_OK) goto bad;   // the bucketed machinery's scratch: partition (every table), pending records (tables with an optimizer); adds to workspace_bytes
    if (hipHostMalloc((void**)&t->h_ctr, sizeof(Counters)) != hipSuccess || hipHostMalloc((void**)&t->h_op, sizeof(OpCounters)) != hipSuccess) {
        rc = fail(MEE_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        goto bad;
    }
    {
        hipError_t e = hipSuccess;
        fill_i64_kernel<<<2048, 256, 0, 0>>>(t->keys, t->capacity, kEmpty);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && t->hits) e = hipMemsetAsync(t->hits, 0, t->capacity * sizeof(uint32_t), 0);
        if (e == hipSuccess && t->sketch) e = hipMemsetAsync(t->sketch, 0, 3ull * sizeof(uint32_t) << t->sketch_log2w, 0);
        if (e == hipSuccess) e = hipMemsetAsync(t->g.ent, 0, S * 16, 0);
        if (e == hipSuccess) e = hipMemsetAsync(t->g.sres, 0, S * 8, 0);
        if (e == hipSuccess && t->bs.gacc) e = hipMemsetAsync(t->bs.gacc, 0, t->max_part * (uint64_t)t->dim * sizeof(double), 0);
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
    else if (!strcmp(name, "find_block")) t->find_block = value;
    else if (!strcmp(name, "prepare_debug")) t->prepare_debug = value;
    else if (!strcmp(name, "find_nt")) t->find_nt = value;
    else if (!strcmp(name, "apply_bucket_max")) t->bk.bucket_max = value > 0 && value <= 352 ? (uint32_t)value : 0u;
    else if (!strcmp(name, "apply_skew_adapt")) t->bk.skew_adapt = t->bk_dd.skew_adapt = value != 0;
    else if (!strcmp(name, "apply_xcd_split")) t->bk.xcd_split = value < 0 ? xcd_split_for_device(t->device) : value < 1024 ? (uint32_t)value : 0u;
    else if (!strcmp(name, "apply_kernel")) t->bk.kernel_choice = value < 0 ? -1 : value != 0;
    else return fail(MEE_ERR_INVALID_ARG, "mee_set_tuning: unknown knob '%s'", name);
    return MEE_OK;
}

int mee_clear(mee_table* t, void* stream) {
    MEE_RANGE("mee_clear");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_clear: null table");
    DeviceGuard g(t->device);
    ++t->handle_epoch;
    fill_i64_kernel<<<2048, 256, 0, as_stream(stream)>>>(t->keys, t->capacity, kEmpty);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// `skip` (nullable): positions whose byte is non-zero take no part (they were served by another table of a tiered pair)
static int upsert_common(mee_table* t, float* plane, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found,
                         void* stream, bool claim, const char* name, const uint8_t* skip = nullptr) {
    if (!t || !plane || (n && (!d_keys || !d_values))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (int rc = check_batch(t, n, name, stream)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    const unsigned gl = grid_for(n, 256, 1u << 22);
    if (claim) {   // insert: creators write at once, an election only among positions that found their key present (see insert_direct_kernel)
        next_epoch(t, st);
        long long* slotof = t->g.sres;   // S >= 2 * max_batch entries: lent as the per-position slot list (as mee_remove does)
        const unsigned gd = grid_for(n, 32, 1u << 16);
#define DIRECT(D4) insert_direct_kernel<D4><<<gd, 256, 0, st>>>(t->keys, (float4*)plane, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4, d_keys, (const float4*)d_values, \
                                                               nn, skip, t->bs.fmask, slotof, t->bs.hidx, t->optimizer, t->init_acc, t->ctr, t->hits, t->epoch)
        if (t->dim4 == 16) DIRECT(16); else if (t->dim4 == 32) DIRECT(32); else DIRECT(0);
#undef DIRECT
        group_last_kernel<<<gl, 256, 0, st>>>(d_keys, nn, t->g, t->bs, t->ctr, t->bs.fmask, t->epoch, &t->ctr->election);
        insert_join_kernel<<<grid_for(n, 256, 2048), 256, 0, st>>>(d_keys, nn, t->bs.fmask, slotof, t->g, t->bs.hidx, t->ctr, t->epoch);
        insert_settle_kernel<<<grid_for(n, 16, 2048), 256, 0, st>>>((float4*)plane, t->dim4, (const float4*)d_values, nn, slotof, t->bs.hidx, t->g, t->ctr, t->epoch);
        MEE_HIP(hipGetLastError());
        return MEE_OK;
    }
    // assign: the bucketed machinery (meepo_dedup.hip: partition by hash bucket, then one kernel — block-local LDS election, one probe and
    // one row copy per distinct key, the found byte to every occurrence)
    return bucket_assign(t, plane, d_keys, d_values, nn, d_found, st);
}

int mee_insert(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, void* stream) {
    MEE_RANGE("mee_insert");
    return upsert_common(t, t ? t->values : nullptr, d_keys, d_values, n, nullptr, stream, true, "mee_insert");
}
int mee_insert_missing(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, const uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_insert_missing");
    if (!d_found && n) return fail(MEE_ERR_INVALID_ARG, "mee_insert_missing: null found mask");
    return upsert_common(t, t ? t->values : nullptr, d_keys, d_values, n, nullptr, stream, true, "mee_insert_missing", d_found);
}
int mee_assign(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_assign");
    return upsert_common(t, t ? t->values : nullptr, d_keys, d_values, n, d_found, stream, false, "mee_assign");
}
int mee_assign_plane(mee_table* t, uint32_t plane, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found,
                     void* stream) {
    MEE_RANGE("mee_assign_plane");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_assign_plane: null table");
    float* p = const_cast<float*>(plane_of(t, plane));
    if (!p) return fail(MEE_ERR_UNSUPPORTED, "mee_assign_plane: plane %u does not exist (optimizer=%u)", plane, t->optimizer);
    return upsert_common(t, p, d_keys, d_values, n, d_found, stream, false, "mee_assign_plane");
}

int mee_remove(mee_table* t, const int64_t* d_keys, size_t n, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_remove");
    if (!t || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_remove: null argument");
    if (int rc = check_batch(t, n, "mee_remove", stream)) return rc;   // borrows group-table scratch: a pending prepared apply is dropped
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    long long* slots = t->g.sres;  // S >= 2 * max_batch entries: reused as the per-position slot list
    ++t->handle_epoch;   // rows are freed: slot handles handed out before this call are stale
    remove_locate_kernel<4><<<grid_for(n, 64, 1u << 16), 256, 0, st>>>(t->keys, t->nb, d_keys, nn, slots, d_found, t->ctr);
    remove_mark_kernel<<<grid_for(n, 256, 1u << 22), 256, 0, st>>>(t->keys, slots, nn);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// shared by mee_find_or_insert (own find pass) and mee_find_or_insert_missing (mask supplied by the caller)
static int find_or_insert_common(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream,
                                 bool own_find_pass, const char* name, int64_t* d_slots_out = nullptr) {
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (int rc = check_batch(t, n, name, stream)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    // pass 1: a plain find serves every key that is already stored (the steady state of training) at find speed and
    // yields the "present before the call" mask; pass 2 runs the insert machinery over the missing positions only.
    uint8_t* fmask = d_found ? d_found : t->bs.fmask;
    if (own_find_pass)
        if (int rc = find_plane(t, t->values, t->default_value, d_keys, n, d_out, fmask, stream, false, false, false, d_slots_out)) return rc;
    // pass 2: every position the mask leaves missing claims its key (or meets the occurrence that did) and writes the key's initial row
    // into the table (creator) and into d_out (everybody): nothing is left for a third pass
    ensure_direct_kernel<64><<<grid_for(n, 256, 8192), 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4,
                                                                d_keys, nn, fmask, t->optimizer, t->init_acc, t->initializer, t->init_scale,
                                                                t->init_seed, t->default_value, t->ctr, t->hits, (float4*)d_out, (long long*)d_slots_out, (long long)handle_tag_of(t));
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_find_or_insert_located(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream) {
    MEE_RANGE("mee_find_or_insert_located");
    if (n && !d_slots_out) return fail(MEE_ERR_INVALID_ARG, "mee_find_or_insert_located: null argument");
    return find_or_insert_common(t, d_keys, n, d_out, d_found, stream, true, "mee_find_or_insert_located", d_slots_out);
}
int mee_find_or_insert(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find_or_insert");
    return find_or_insert_common(t, d_keys, n, d_out, d_found, stream, true, "mee_find_or_insert");
}
int mee_find_or_insert_admit(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, uint32_t min_count, void* stream) {
    MEE_RANGE("mee_find_or_insert_admit");
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_or_insert_admit: null argument");
    if (!t->sketch) return fail(MEE_ERR_UNSUPPORTED, "mee_find_or_insert_admit: table was created without MEE_FLAG_ADMISSION");
    if (int rc = check_batch(t, n, "mee_find_or_insert_admit", stream)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    uint8_t* fmask = d_found ? d_found : t->bs.fmask;          // present before the call
    uint8_t* skip = reinterpret_cast<uint8_t*>(t->bs.occ);     // max_batch x 4 bytes of scratch no other step of this call uses
    if (int rc = find_plane(t, t->values, t->default_value, d_keys, n, d_out, fmask, stream)) return rc;
    const unsigned gl = grid_for(n, 256, 4096);
    sketch_add_kernel<<<gl, 256, 0, st>>>(d_keys, fmask, n, t->sketch, t->sketch_log2w);
    sketch_decide_kernel<<<gl, 256, 0, st>>>(d_keys, fmask, n, t->sketch, t->sketch_log2w, min_count, skip);
    ensure_direct_kernel<64><<<grid_for(n, 256, 8192), 256, 0, st>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4,
                                                                d_keys, (uint32_t)n, skip, t->optimizer, t->init_acc, t->initializer, t->init_scale,
                                                                t->init_seed, t->default_value, t->ctr, t->hits, (float4*)d_out);   // admitted positions get their initial row here
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}
int mee_admission_decay(mee_table* t, uint32_t shift, void* stream) {
    MEE_RANGE("mee_admission_decay");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_admission_decay: null table");
    if (!t->sketch) return fail(MEE_ERR_UNSUPPORTED, "mee_admission_decay: table was created without MEE_FLAG_ADMISSION");
    DeviceGuard g(t->device);
    sketch_decay_kernel<<<2048, 256, 0, as_stream(stream)>>>(t->sketch, 3ull << t->sketch_log2w, shift);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_find_or_insert_missing(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, const uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find_or_insert_missing");
    if (!d_found && n) return fail(MEE_ERR_INVALID_ARG, "mee_find_or_insert_missing: null found mask");
    return find_or_insert_common(t, d_keys, n, d_out, const_cast<uint8_t*>(d_found), stream, false, "mee_find_or_insert_missing");
}

int mee_locate(const mee_table* t, const int64_t* d_keys, size_t n, int64_t* d_slots_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_locate");
    if (!t || (n && (!d_keys || !d_slots_out))) return fail(MEE_ERR_INVALID_ARG, "mee_locate: null argument");
    if (n > 0xFFFFFFFFull) return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_locate: n=%zu exceeds 2^32 - 1", n);
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    remove_locate_kernel<4><<<grid_for(n, 64, 1u << 16), 256, 0, as_stream(stream)>>>(t->keys, t->nb, d_keys, (uint32_t)n, (long long*)d_slots_out, d_found, t->ctr);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// group the batch's keys and plan the duplicate reduction (everything that does not need the grads)
// One sparse-optimizer step: the bucketed apply (meepo_apply.hip) — partition by hash bucket (skipped after mee_apply_prepare / a training
// forward that carried it), then ONE kernel of block-local LDS dedup + update.  (Rounds 1-3 kept a second implementation beside it, a global
// group table with five launches per step; it is gone: batches of any size up to max_batch take this path.)
// `d_slots` (nullable): the slot of every position as mee_find_located of the same step reported it.
static int apply_common(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, const OptArgs& a, void* stream,
                        const char* name, const uint32_t* d_gidx = nullptr, const int64_t* d_slots = nullptr) {
    if (!t || (n && (!d_keys || !d_grads))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (t->optimizer != a.kind) return fail(MEE_ERR_UNSUPPORTED, "%s: table was created with optimizer=%u", name, t->optimizer);
    if (int rc = check_batch(t, n, name, stream, false)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const uint32_t nn = (uint32_t)n;
    if (t->prepared_n) {  // the grad-independent half was done ahead of time (mee_apply_prepare), possibly on another stream
        if (t->prepared_n != n || t->prepared_keys != d_keys)
            return fail(MEE_ERR_INVALID_ARG, "%s: keys/n differ from the pending mee_apply_prepare", name);
        t->prepared_n = 0; t->prepared_keys = nullptr;
    } else if (int rc = bucket_apply_prepare(t, d_keys, nn, st)) return rc;
    return bucket_apply_launch(t, d_grads, nn, a, d_gidx, d_slots, st);
}

static OptArgs adam_args(float lr, float beta1, float beta2, float eps, uint64_t step);

// ---- grouped apply: ONE sparse-optimizer step over the jagged batch of a whole group (meepo_group.hip holds the group) ----
// locate (member << 48 | slot per position) -> the ordinary group / plan passes with the located rows as "keys" (two
// occurrences of a key of one table are the same row; keys of different tables never collide) -> the three apply passes
// with the probe replaced by decoding the located row.  The group's scratch table lends the group table / lists / counters.
static int group_apply_common(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n,
                              const OptArgs& a, void* stream, const char* name, uint64_t off_stride = 1, const uint32_t* d_gidx = nullptr,
                              const int64_t* d_located = nullptr) {
    if (!g || (!d_located && !d_offsets) || (n && ((!d_keys && !d_located) || !d_grads))) return fail(MEE_ERR_INVALID_ARG, "%s: null argument", name);
    if (!g->scratch) return fail(MEE_ERR_UNSUPPORTED, "%s: group was created with max_apply_batch = 0 or its tables have no optimizer", name);
    if (g->optimizer != a.kind) return fail(MEE_ERR_UNSUPPORTED, "%s: the group's tables were created with optimizer=%u", name, g->optimizer);
    if (n > g->max_apply_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "%s: n=%zu exceeds the group's max_apply_batch=%llu", name, n, (unsigned long long)g->max_apply_batch);
    if (n == 0) return MEE_OK;
    if (int rc = group_refresh(g, stream)) return rc;
    DeviceGuard guard(g->device);
    hipStream_t st = as_stream(stream);
    mee_table* t = g->scratch;
    const uint32_t nn = (uint32_t)n;
    // the located rows of the batch: handed over by the forward lookup of the same step, or found by a probe pass of our own
    const int64_t* gslot = d_located;
    if (!gslot) {
        if (int rc = group_locate(g, d_keys, d_offsets, off_stride, n, g->d_gslot, st)) return rc;
        gslot = g->d_gslot;
    }
    // the located rows are perfect keys (member << 48 | slot; absent positions EMPTY): the bucketed apply takes them as the batch's keys AND as
    // its slot handles — partition, then one dedup + update kernel whose work items fetch their member's planes from the descriptors (r1-r2: a
    // plan pass over a group table + three kernels; 5 launches, 26 tables x 8192 keys 158 us)
    if (int rc = bucket_apply_prepare(t, gslot, nn, st)) return rc;
    return bucket_apply_launch(t, d_grads, nn, a, d_gidx, gslot, st, g->d_desc, g->n_tables);
}

int mee_group_apply_adagrad(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr,
                            float eps, void* stream) {
    MEE_RANGE("mee_group_apply_adagrad");
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps;
    return group_apply_common(g, d_keys, d_offsets, d_grads, n, a, stream, "mee_group_apply_adagrad");
}
int mee_group_apply_adam(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr,
                         float beta1, float beta2, float eps, uint64_t step, void* stream) {
    MEE_RANGE("mee_group_apply_adam");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_group_apply_adam: step must be >= 1");
    OptArgs a{};
    a.kind = MEE_OPT_ADAM; a.eps = eps;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr * sqrt(bc2) / bc1);  // SPEC.md §4
    a.omb1 = 1.0f - beta1; a.omb2 = 1.0f - beta2;
    return group_apply_common(g, d_keys, d_offsets, d_grads, n, a, stream, "mee_group_apply_adam");
}

int mee_group_apply_adagrad_pooled(mee_group* g, const int64_t* d_keys, const uint64_t* d_bag_offsets, size_t bags_per_table,
                                   const float* d_bag_grads, const uint32_t* d_grad_index, const int64_t* d_located, size_t n, float lr,
                                   float eps, void* stream) {
    MEE_RANGE("mee_group_apply_adagrad_pooled");
    if (n && (!d_grad_index || !bags_per_table)) return fail(MEE_ERR_INVALID_ARG, "mee_group_apply_adagrad_pooled: null index / zero bags_per_table");
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps;
    a.grad_rows = g ? (uint32_t)(g->n_tables * bags_per_table) : 0;   // one grad row per bag
    // the members' key segments are bounded by every bags_per_table-th bag offset
    return group_apply_common(g, d_keys, d_bag_offsets, d_bag_grads, n, a, stream, "mee_group_apply_adagrad_pooled", bags_per_table, d_grad_index,
                              d_located);
}
int mee_group_apply_adam_pooled(mee_group* g, const int64_t* d_keys, const uint64_t* d_bag_offsets, size_t bags_per_table,
                                const float* d_bag_grads, const uint32_t* d_grad_index, const int64_t* d_located, size_t n, float lr,
                                float beta1, float beta2, float eps, uint64_t step, void* stream) {
    MEE_RANGE("mee_group_apply_adam_pooled");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_group_apply_adam_pooled: step must be >= 1");
    if (n && (!d_grad_index || !bags_per_table)) return fail(MEE_ERR_INVALID_ARG, "mee_group_apply_adam_pooled: null index / zero bags_per_table");
    OptArgs a = adam_args(lr, beta1, beta2, eps, step);
    a.grad_rows = g ? (uint32_t)(g->n_tables * bags_per_table) : 0;   // one grad row per bag
    return group_apply_common(g, d_keys, d_bag_offsets, d_bag_grads, n, a, stream, "mee_group_apply_adam_pooled", bags_per_table, d_grad_index,
                              d_located);
}

int mee_apply_prepare(mee_table* t, const int64_t* d_keys, size_t n, void* stream) {
    MEE_RANGE("mee_apply_prepare");
    if (!t || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_apply_prepare: null argument");
    if (t->optimizer == MEE_OPT_NONE) return fail(MEE_ERR_UNSUPPORTED, "mee_apply_prepare: table has no optimizer");
    if (t->prepared_n) return fail(MEE_ERR_INVALID_ARG, "mee_apply_prepare: a prepared apply is already pending");
    if (n > t->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_apply_prepare: n=%zu exceeds config.max_batch=%llu", n, (unsigned long long)t->max_batch);
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    if (int rc = bucket_apply_prepare(t, d_keys, (uint32_t)n, as_stream(stream))) return rc;   // the partition half of the apply: positions and keys in bucket order
    t->prepared_n = n; t->prepared_keys = d_keys; t->prepared_by_forward = false;
    return MEE_OK;
}

int mee_find_or_insert_located_prepare(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream) {
    MEE_RANGE("mee_find_or_insert_located_prepare");
    if (!t || (n && (!d_keys || !d_out || !d_slots_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_or_insert_located_prepare: null argument");
    if (t->optimizer == MEE_OPT_NONE) return mee_find_or_insert_located(t, d_keys, n, d_out, d_found, d_slots_out, stream);
    if (int rc = check_batch(t, n, "mee_find_or_insert_located_prepare", stream, true, false)) return rc;   // (refuses while another prepared apply is pending)
    if (n == 0) return MEE_OK;
    uint8_t* fmask = d_found ? d_found : t->bs.fmask;
    // pass 1: the training forward's launch (located find of the stored keys + the partition of the backward); pass 2: the missing positions
    // claim their keys and write the initial rows (table, d_out) and the slot handles, as in mee_find_or_insert_located — it touches none of the
    // scratch the partition left for the apply, and the partition depends on the batch's keys alone
    if (int rc = mee_find_located_prepare(t, d_keys, n, d_out, fmask, d_slots_out, stream)) return rc;
    DeviceGuard g(t->device);
    ensure_direct_kernel<64><<<grid_for(n, 256, 8192), 256, 0, as_stream(stream)>>>(t->keys, (float4*)t->values, (float4*)t->s1, (float4*)t->s2, t->nb, t->dim4,
                                                                d_keys, (uint32_t)n, fmask, t->optimizer, t->init_acc, t->initializer, t->init_scale,
                                                                t->init_seed, t->default_value, t->ctr, t->hits, (float4*)d_out, (long long*)d_slots_out, (long long)handle_tag_of(t));
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_apply_discard(mee_table* t, void* stream) {
    MEE_RANGE("mee_apply_discard");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_apply_discard: null table");
    if (!t->prepared_n) return MEE_OK;
    DeviceGuard g(t->device);
    const uint32_t nn = (uint32_t)t->prepared_n;
    (void)nn;
    if (int rc = bucket_apply_discard(t, as_stream(stream))) return rc;   // (counts the partition as consumed; it leaves nothing else behind)
    t->prepared_n = 0; t->prepared_keys = nullptr;
    return MEE_OK;
}

int mee_apply_adagrad(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps, void* stream) {
    MEE_RANGE("mee_apply_adagrad");
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adagrad");
}
static OptArgs adam_args(float lr, float beta1, float beta2, float eps, uint64_t step) {
    OptArgs a{};
    a.kind = MEE_OPT_ADAM; a.eps = eps;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr * sqrt(bc2) / bc1);  // SPEC.md §4
    a.omb1 = 1.0f - beta1; a.omb2 = 1.0f - beta2;
    return a;
}
int mee_apply_adam(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1, float beta2,
                   float eps, uint64_t step, void* stream) {
    MEE_RANGE("mee_apply_adam");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adam: step must be >= 1");
    return apply_common(t, d_keys, d_grads, n, adam_args(lr, beta1, beta2, eps, step), stream, "mee_apply_adam");
}
// the same with the slots the forward lookup of this step located (mee_find_located): the main pass skips its probe
int mee_apply_adagrad_located(mee_table* t, const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr, float eps,
                              void* stream) {
    MEE_RANGE("mee_apply_adagrad_located");
    if (n && !d_slots) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adagrad_located: null slots");
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adagrad_located", nullptr, d_slots);
}
int mee_apply_adam_located(mee_table* t, const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr, float beta1,
                           float beta2, float eps, uint64_t step, void* stream) {
    MEE_RANGE("mee_apply_adam_located");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adam_located: step must be >= 1");
    if (n && !d_slots) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adam_located: null slots");
    return apply_common(t, d_keys, d_grads, n, adam_args(lr, beta1, beta2, eps, step), stream, "mee_apply_adam_located", nullptr, d_slots);
}
// the grad of position i is row d_grad_index[i] of d_grads (pooled lookups: the bag's grad row serves every key of the bag)
static int check_grad_rows(size_t n, const uint32_t* d_grad_index, size_t n_grad_rows, const char* name) {
    if (n && (!d_grad_index || n_grad_rows == 0 || n_grad_rows > 0xFFFFFFFFull))
        return fail(MEE_ERR_INVALID_ARG, "%s: null index, or n_grad_rows not in [1, 2^32)", name);
    return MEE_OK;
}
int mee_apply_adagrad_indexed(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index,
                              size_t n, float lr, float eps, void* stream) {
    MEE_RANGE("mee_apply_adagrad_indexed");
    if (int rc = check_grad_rows(n, d_grad_index, n_grad_rows, "mee_apply_adagrad_indexed")) return rc;
    OptArgs a{};
    a.kind = MEE_OPT_ADAGRAD; a.lr = lr; a.eps = eps; a.grad_rows = (uint32_t)n_grad_rows;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adagrad_indexed", d_grad_index);
}
int mee_apply_adam_indexed(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index,
                           size_t n, float lr, float beta1, float beta2, float eps, uint64_t step, void* stream) {
    MEE_RANGE("mee_apply_adam_indexed");
    if (step == 0) return fail(MEE_ERR_INVALID_ARG, "mee_apply_adam_indexed: step must be >= 1");
    if (int rc = check_grad_rows(n, d_grad_index, n_grad_rows, "mee_apply_adam_indexed")) return rc;
    OptArgs a = adam_args(lr, beta1, beta2, eps, step);
    a.grad_rows = (uint32_t)n_grad_rows;
    return apply_common(t, d_keys, d_grads, n, a, stream, "mee_apply_adam_indexed", d_grad_index);
}

/* duplicate-key reduction with row sums, sync-free (meepo_dedup.hip): see include/meepo_embedding.h */
int mee_dedup_sum(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, int64_t* d_uniq_out, float* d_gsum_out,
                  uint32_t* d_counts_out, int64_t* d_inverse_out, int64_t miss_index, void* stream) {
    MEE_RANGE("mee_dedup_sum");
    if (!t || (n && (!d_keys || !d_uniq_out))) return fail(MEE_ERR_INVALID_ARG, "mee_dedup_sum: null argument");
    if ((d_grads == nullptr) != (d_gsum_out == nullptr)) return fail(MEE_ERR_INVALID_ARG, "mee_dedup_sum: d_grads and d_gsum_out go together (both or neither)");
    if (int rc = check_batch(t, n, "mee_dedup_sum", stream)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    return bucket_dedup_sum(t, d_keys, d_grads, (uint32_t)n, d_uniq_out, d_gsum_out, d_counts_out, d_inverse_out, miss_index, as_stream(stream));
}

/* sync-free duplicate elimination of a key batch: d_uniq_out[n] = the distinct non-reserved keys (unspecified order) with EMPTY
 * padding, d_inverse_out[i] = index of keys[i] in d_uniq_out, or miss_index for reserved keys.  Nothing returns to the host. */
int mee_dedup_keys(mee_table* t, const int64_t* d_keys, size_t n, int64_t* d_uniq_out, int64_t* d_inverse_out, int64_t miss_index,
                   void* stream) {
    MEE_RANGE("mee_dedup_keys");
    if (!t || (n && (!d_keys || !d_uniq_out || !d_inverse_out))) return fail(MEE_ERR_INVALID_ARG, "mee_dedup_keys: null argument");
    if (int rc = check_batch(t, n, "mee_dedup_keys", stream)) return rc;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    return bucket_dedup_keys(t, d_keys, (uint32_t)n, d_uniq_out, d_inverse_out, miss_index, as_stream(stream));
}

}  // extern "C"


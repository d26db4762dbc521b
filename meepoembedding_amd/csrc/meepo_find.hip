// meepo_find.hip — the lookups of the table: find (the headline kernel), the training forward (find + the apply's partition in one
// launch), find_many, the sparse second pass, pooled find; their half of the C-ABI.  Table layout and scratch: meepo_table.hip,
// meepo_table_int.h.  Reference anchor: /root/reference/README.md:2 (no code in the snapshot: semantics from SPEC.md §3).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "meepo_apply_part.h"


namespace mee {

// ---- find (SPEC.md §3) — the headline kernel --------------------------------------------------------------
// One tile per key, R keys in flight per tile: the R bucket lines are requested back to back, then the R rows.
// DIM4 = dim/4 when it is a multiple of 16 (each lane moves DIM4/16 float4 per row), 0 = any dim at run time.

// the find of n positions by `n_waves` waves of which this is wave `wave` (each wave step takes 4R consecutive positions)
template <int DIM4, int R, int NT>
__device__ __forceinline__ void find_span(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values, uint64_t nb,
                                          const int64_t* __restrict__ keys, uint64_t n, f32x4* __restrict__ out,
                                          uint8_t* __restrict__ found, float defv, uint32_t dim4_rt, uint32_t* hits,
                                          int64_t* __restrict__ slots_out, uint64_t wave, uint64_t n_waves, int64_t handle_tag = 0) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    constexpr int KPW = 4 * R;
    const f32x4 def4 = {defv, defv, defv, defv};

    for (uint64_t base = wave * KPW; base < n; base += n_waves * KPW) {
        int64_t key[R];
        int64_t slot[R];
        uint64_t b[R];
        int64_t kb[R];
        bool inb[R], act[R];
        // ONE coalesced load brings the wave step's 4R keys (lane j reads keys[base + j]); the tiles take theirs by shuffle.  (A load per
        // tile and round made the compiler wait for round r's key before it requested round r + 1's: a dependent memory round trip per round.)
        const int64_t kmine = (lane < KPW && base + lane < n) ? keys[base + lane] : kEmpty;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = base + r * 4 + tile;
            inb[r] = i < n;
            key[r] = __shfl(kmine, r * 4 + tile);
            act[r] = inb[r] && !reserved_key(key[r]);
            if constexpr ((NT & 8) != 0) {  // second-tier pass: only positions an earlier find left as missing
                inb[r] = act[r] = act[r] && found[i] == 0;
            }
            if constexpr ((NT & 128) != 0) {  // owner side of a padded sharded exchange: EMPTY positions are padding nobody reads — no row; their
                // found byte says "served" so that a second-tier pass over the same buffers (mee_find_missing / mee_find_or_insert_missing on the
                // cold table of a tiered shard) leaves them alone instead of reading a byte nobody wrote
                if (inb[r] && !act[r] && tl == 0 && found) found[i] = 1;
                inb[r] = act[r];
            }
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
                    else bb = next_bucket(bb, step_of(key[r], nb), nb);
                }
                if (!__any(pend)) break;
                k = pend ? tkeys[bb * kW + tl] : kEmpty;
            }
        }
        if constexpr ((NT & 16) != 0) {  // access statistics for the hot/cold policy (sampled calls only)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (slot[r] >= 0 && tl == 0) atomicAdd(&hits[slot[r]], 1u);
        }
        if constexpr (DIM4 != 0) {
            constexpr int C = DIM4 / 16;
            f32x4 row[R][C];
            // Common case, decided per wave: every position of the wave step is inside the batch and was found.  Then the R row loads
            // and the R stores are straight-line code and leave back to back; with a per-lane condition around each load the compiler
            // waited for round r's row before it requested round r + 1's (seen in the ISA of the located variant: +8 us per 256K keys).
            bool all_hit = true;
#pragma unroll
            for (int r = 0; r < R; ++r) all_hit = all_hit && inb[r] && slot[r] >= 0;
            if (__all(all_hit)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        row[r][c] = (NT & 1) ? __builtin_nontemporal_load(&values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]) : values[(uint64_t)slot[r] * DIM4 + c * 16 + tl];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint64_t i = base + r * 4 + tile;
#pragma unroll
                    for (int c = 0; c < C; ++c) { if (NT & 4) out[i * DIM4 + c * 16 + tl] = row[r][c]; else __builtin_nontemporal_store(row[r][c], &out[i * DIM4 + c * 16 + tl]); }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        row[r][c] = slot[r] >= 0 ? ((NT & 1) ? __builtin_nontemporal_load(&values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]) : values[(uint64_t)slot[r] * DIM4 + c * 16 + tl]) : def4;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint64_t i = base + r * 4 + tile;
                    if (inb[r] && (!(NT & 8) || slot[r] >= 0)) {
#pragma unroll
                        for (int c = 0; c < C; ++c) { if (NT & 4) out[i * DIM4 + c * 16 + tl] = row[r][c]; else __builtin_nontemporal_store(row[r][c], &out[i * DIM4 + c * 16 + tl]); }
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i = base + r * 4 + tile;
                if (inb[r] && (!(NT & 8) || slot[r] >= 0))
                    for (uint32_t c = tl; c < dim4; c += 16)
                        out[i * dim4 + c] = slot[r] >= 0 ? values[(uint64_t)slot[r] * dim4 + c] : def4;
            }
        }
        if constexpr ((NT & 64) != 0) {  // mee_find_located: the slot of every position (-1 = absent), for the apply of the same step
            // lane j < 4R collects the slot of position base + j (round j / 4, tile j % 4): ONE coalesced store per wave step
            int64_t mine = -1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t v = __shfl(slot[r], (lane & 3) * kW);
                if ((lane >> 2) == r) mine = v;
            }
            if (lane < KPW && base + lane < n) slots_out[base + lane] = mine >= 0 ? (mine | handle_tag) : mine;   // tag: the table's layout epoch (see handle_tag_of)
        }
        if (found && !(NT & 32)) {  // NT&32: rows only (last pass of find_or_insert: found keeps meaning "present before")
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t i0 = base + r * 4;
                if ((NT & (8 | 128)) == 0 && i0 + 4 <= n && (reinterpret_cast<uintptr_t>(found) & 3) == 0) {
                    // the four tiles' found bytes of this round leave as ONE aligned 4-byte store
                    const uint64_t m = __ballot(slot[r] >= 0);
                    const uint32_t w = (uint32_t)(m & 1) | ((uint32_t)((m >> 16) & 1) << 8) | ((uint32_t)((m >> 32) & 1) << 16) |
                                       ((uint32_t)((m >> 48) & 1) << 24);
                    if (lane == 0) *reinterpret_cast<uint32_t*>(found + i0) = w;
                } else {
                    const uint64_t i = i0 + tile;
                    if (inb[r] && tl == 0 && (!(NT & 8) || slot[r] >= 0)) found[i] = slot[r] >= 0;   // (NT & 128: inb excludes padding)
                }
            }
        }
    }
}

#ifndef MEE_FIND_TIMELINE
#define MEE_FIND_TIMELINE 0
#endif
#if MEE_FIND_TIMELINE
__device__ unsigned long long* g_find_dbg = nullptr;
#endif
template <int DIM4, int R, int NT>
__global__ __launch_bounds__(256) void find_kernel(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values,
                                                   uint64_t nb, const int64_t* __restrict__ keys, uint64_t n,
                                                   f32x4* __restrict__ out, uint8_t* __restrict__ found, float defv,
                                                   uint32_t dim4_rt, uint32_t* hits, int64_t* __restrict__ slots_out = nullptr, int64_t handle_tag = 0) {
#if MEE_FIND_TIMELINE   // diagnostic builds only (tools/find_timeline.py): wave 0 of every block stamps its start, its end and the XCD it ran on
    unsigned long long t0_ = 0;
    if (threadIdx.x == 0) t0_ = wall_clock64();
#endif
    find_span<DIM4, R, NT>(tkeys, values, nb, keys, n, out, found, defv, dim4_rt, hits, slots_out,
                           (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), (uint64_t)gridDim.x * (blockDim.x >> 6), handle_tag);
#if MEE_FIND_TIMELINE
    if (threadIdx.x == 0 && g_find_dbg && blockIdx.x < 16384) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_find_dbg[blockIdx.x * 4 + 0] = t0_; g_find_dbg[blockIdx.x * 4 + 1] = wall_clock64(); g_find_dbg[blockIdx.x * 4 + 2] = xcc & 0xf;
    }
#endif
}

// The training forward (mee_find_located_prepare): the located find whose launch gives its first `part_blocks` blocks the partition role of
// the bucketed apply (meepo_apply_part.h).  The partition of the step's backward — a latency-bound 15-18 us of LDS histograms for a 256K-key
// batch — runs beside the forward's row gather, which is bound by bytes and takes twice as long: the backward starts with its update kernel.
// Block size = the find's own 256 threads.  The first version used 1024-thread blocks for the sake of the partition role (64 blocks x 1024
// threads); measured with the role switched off, the FIND in 1024-thread blocks takes 47 us against 39 us in 256-thread blocks (a block's 16
// waves each do one short pass, the block holds its 16 wave slots until the slowest of them is done; 512-thread blocks: 42 us) — the launch
// took as long as its slow find and hid nothing.  With 256-thread blocks the role runs as 128 blocks x 256 threads x 8 keys, each making two
// round trips to memory (meepo_apply_part.h): 42.5 us for the launch against 40 us with the role switched off.
constexpr int kFindPrepareThreads = 256;
constexpr int kFindPrepareR = 2;   // keys in flight per tile at dim 64
template <int DIM4, int R, int NT>
__global__ __launch_bounds__(kFindPrepareThreads, 8) void find_prepare_kernel(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values, uint64_t nb,
                                                           const int64_t* __restrict__ keys, uint64_t n, f32x4* __restrict__ out,
                                                           uint8_t* __restrict__ found, float defv, uint32_t dim4_rt, int64_t* __restrict__ slots_out,
                                                           int64_t handle_tag, uint32_t part_blocks, uint32_t nbk_hash, uint32_t nbk, uint32_t per_block,
                                                           BucketScratch bk, uint32_t* status, OpCounters* op, uint32_t xcd_split) {
    extern __shared__ unsigned long long part_lds[];   // PartHot, then one counter per bucket (meepo_apply_part.h)
    __shared__ unsigned long long part_wsum[kFindPrepareThreads / 64];
#if MEE_FIND_TIMELINE   // diagnostic builds only (tools/prepare_timeline.py): thread 0 of every block stamps its start, its end and its role
    unsigned long long t0_ = 0;
    if (threadIdx.x == 0) t0_ = wall_clock64();
#endif
    if (blockIdx.x < part_blocks) {   // block-uniform
        PartHot* hot = reinterpret_cast<PartHot*>(part_lds);
        sort_role<kFindPrepareThreads>(keys, (uint32_t)n, nbk_hash, nbk, per_block, blockIdx.x, part_blocks, bk, status, op, reinterpret_cast<uint32_t*>(hot + 1), part_wsum, hot, true, xcd_split);
#if MEE_FIND_TIMELINE
        if (threadIdx.x == 0 && g_find_dbg && blockIdx.x < 16384) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g_find_dbg[blockIdx.x * 4 + 0] = t0_; g_find_dbg[blockIdx.x * 4 + 1] = wall_clock64(); g_find_dbg[blockIdx.x * 4 + 2] = 100;
        }
#endif
        return;
    }
    find_span<DIM4, R, NT>(tkeys, values, nb, keys, n, out, found, defv, dim4_rt, nullptr, slots_out,
                           (uint64_t)(blockIdx.x - part_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6), (uint64_t)(gridDim.x - part_blocks) * (blockDim.x >> 6),
                           handle_tag);
#if MEE_FIND_TIMELINE
    if (threadIdx.x == 0 && g_find_dbg && blockIdx.x < 16384) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_find_dbg[blockIdx.x * 4 + 0] = t0_; g_find_dbg[blockIdx.x * 4 + 1] = wall_clock64(); g_find_dbg[blockIdx.x * 4 + 2] = 1;
    }
#endif
}

// Several lookup requests of one table in ONE launch (mee_find_many): the per-launch latency floor (~5 us: dispatch + the dependent
// chain of the first and last waves) is paid once for all of them, so four queued 256K-key requests run at the rate of one 1M-key
// launch.  The request descriptors travel in the kernel argument; a block finds its request by its index (<= kMaxFindRequests entries).
constexpr int kMaxFindRequests = 16;
struct FindMany {
    const int64_t* keys[kMaxFindRequests];
    f32x4* out[kMaxFindRequests];
    uint8_t* found[kMaxFindRequests];
    uint64_t n[kMaxFindRequests];
    uint32_t first_block[kMaxFindRequests + 1];
    uint32_t count;
};
template <int DIM4, int R, int NT>
__global__ __launch_bounds__(256) void find_many_kernel(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values, uint64_t nb,
                                                        FindMany m, float defv, uint32_t dim4_rt) {
    uint32_t q = 0;
    while (q + 1 < m.count && blockIdx.x >= m.first_block[q + 1]) ++q;   // block-uniform
    find_span<DIM4, R, NT>(tkeys, values, nb, m.keys[q], m.n[q], m.out[q], m.found[q], defv, dim4_rt, nullptr, nullptr,
                           (uint64_t)(blockIdx.x - m.first_block[q]) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                           (uint64_t)(m.first_block[q + 1] - m.first_block[q]) * (blockDim.x >> 6));
}

// ---- sparse second pass (SPEC.md §3 find_missing; last pass of find_or_insert): only positions whose found byte is 0 -----
// A wave reads 64 keys + found bytes with one coalesced load each and leaves at once when nothing is missing (the common
// case: a hot tier that holds the working set, a trained vocabulary); the missing ones are probed four at a time.
// FLAGS bit 0: set found[i] = 1 where the key is stored here (tier pass; off = rows only), bit 1: count the hit.
template <int FLAGS>
__global__ __launch_bounds__(256) void find_missing_kernel(const int64_t* __restrict__ tkeys, const float4* __restrict__ values,
                                                           uint64_t nb, uint32_t dim4, const int64_t* __restrict__ keys, uint64_t n,
                                                           float4* __restrict__ out, uint8_t* __restrict__ found, uint32_t* hits) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t base = wave * 64; base < n; base += n_waves * 64) {
        const uint64_t i = base + lane;
        const int64_t k = i < n ? keys[i] : kEmpty;
        uint64_t rest = __ballot(i < n && found[i] == 0 && !reserved_key(k));
        while (rest) {  // wave-uniform
            uint64_t mm = rest;
            int p = -1;
            for (int q = 0; q <= tile; ++q) {
                if (mm) { p = __ffsll((unsigned long long)mm) - 1; mm &= mm - 1; } else p = -1;
            }
            const int64_t key = __shfl(k, p >= 0 ? p : 0);
            bool is_new, full;
            const int64_t slot = tile_locate<false, false>(const_cast<int64_t*>(tkeys), nb, key, p >= 0, tile, tl, is_new, full);
            if (p >= 0 && slot >= 0) {
                const uint64_t dst = (base + (uint64_t)p) * dim4, src = (uint64_t)slot * dim4;
                for (uint32_t c = tl; c < dim4; c += 16) out[dst + c] = values[src + c];
                if (tl == 0) {
                    if (FLAGS & 1) found[base + p] = 1;
                    if (FLAGS & 2) atomicAdd(&hits[slot], 1u);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rest &= rest - 1;
        }
    }
}

// ---- pooled find (SPEC.md §3): out[b,:] = the rows of bag b's keys added up in position order (sum | mean) ----------------
// The partial sum of a bag lives in registers, so a bag costs ONE output row instead of one per key: with L keys per bag
// the write traffic of the lookup drops from 256 B per key to 256/L.  Additions happen in position order (bit-exact
// against a sequential sum).  A wave takes four consecutive bags.  Short bags: one tile per bag, U keys of the bag in
// flight (all bucket lines requested, then all rows).  Bags of kPoolLong keys or more: the four tiles work on ONE bag
// together — tile t probes positions i + 4u + t — and every tile then adds the 4U rows in position order out of the
// other tiles' registers (shuffles), so a long bag has 4U keys in flight instead of U.
constexpr uint32_t kPoolLong = 16;

// probe + row load of up to U keys per tile (the find_kernel pattern); row[u] is only defined where inb[u]
template <int DIM4, int U, int C>
__device__ __forceinline__ void pooled_fetch(const int64_t* __restrict__ tkeys, const float4* __restrict__ values, uint64_t nb,
                                             uint32_t dim4, const int64_t (&key)[U], const uint64_t (&pos)[U],
                                             const bool (&inb)[U], int tile, int tl, float4 def4, float4 (&row)[U][C],
                                             uint8_t* __restrict__ found, int64_t* __restrict__ located = nullptr, uint64_t member = 0) {
    int64_t slot[U], kb[U];
    uint64_t bk[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        act[u] = inb[u] && !reserved_key(key[u]);
        bk[u] = bucket_of(key[u], nb);
        kb[u] = act[u] ? tkeys[bk[u] * kW + tl] : kEmpty;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        slot[u] = -1;
        bool pend = act[u];
        uint64_t bb = bk[u], steps = 0;
        int64_t k = kb[u];
        while (true) {
            const uint32_t tm = tile_bits(__ballot(pend && k == key[u]), tile);
            const uint32_t te = tile_bits(__ballot(pend && k == kEmpty), tile);
            if (pend) {
                if (tm) { slot[u] = (int64_t)(bb * kW) + (__ffs(tm) - 1); pend = false; }
                else if (te || ++steps >= nb) pend = false;
                else bb = next_bucket(bb, step_of(key[u], nb), nb);
            }
            if (!__any(pend)) break;
            k = pend ? tkeys[bb * kW + tl] : kEmpty;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (DIM4 != 0 || (uint32_t)(c * 16 + tl) < dim4)
                row[u][c] = slot[u] >= 0 ? values[(uint64_t)slot[u] * dim4 + c * 16 + tl] : def4;
        if (found && inb[u] && tl == 0) found[pos[u]] = slot[u] >= 0;
        // the located row (member << 48 | slot, EMPTY when absent): lets the backward skip its own probe pass
        if (located && inb[u] && tl == 0) located[pos[u]] = slot[u] >= 0 ? (int64_t)((member << kGroupSlotBits) | (uint64_t)slot[u]) : kEmpty;
    }
}

// BPW = bags per wave: 4 = the hybrid above (batches of mostly short bags), 1 = every bag gets a whole wave (batches whose
// AVERAGE bag is long: a wave that had to walk four long bags one after the other would be latency-bound).
// GROUPED (mee_group_find_pooled): bag b belongs to member table b / bags_per_table; the planes come from its descriptor.
template <int DIM4, int U, int BPW, bool GROUPED = false>
__global__ __launch_bounds__(256) void find_pooled_kernel(const int64_t* __restrict__ tkeys_, const float4* __restrict__ values_,
                                                          uint64_t nb_, const int64_t* __restrict__ keys,
                                                          const uint64_t* __restrict__ offsets, uint64_t n_bags,
                                                          float4* __restrict__ out, uint8_t* __restrict__ found, float defv,
                                                          uint32_t dim4_rt, int mean, const GroupDesc* __restrict__ desc = nullptr,
                                                          uint64_t bags_per_table = 1, int64_t* __restrict__ located = nullptr,
                                                          uint64_t n_keys = ~0ull) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t dim4 = DIM4 ? DIM4 : dim4_rt;
    constexpr int C = DIM4 ? DIM4 / 16 : 16;   // float4 per lane per row (any dim: up to 1024 floats = 16 per lane)
    for (uint64_t b0 = wave * BPW; b0 < n_bags; b0 += n_waves * BPW) {
        const uint64_t bag = BPW == 4 ? b0 + tile : b0;
        const bool has = bag < n_bags;
        uint64_t begin = has ? offsets[bag] : 0, end = has ? offsets[bag + 1] : 0;
        end = end < n_keys ? end : n_keys;          // offsets are the caller's: never read past the key array,
        begin = begin < end ? begin : end;          // and a decreasing pair is an empty bag
        const int64_t* tkeys = tkeys_;
        const float4* values = values_;
        uint64_t nb = nb_;
        float4 def4 = make_float4(defv, defv, defv, defv);
        uint64_t member = 0;
        if constexpr (GROUPED) {
            member = (has ? bag : 0) / bags_per_table;
            const GroupDesc d = desc[member];
            tkeys = d.tkeys; values = d.values; nb = d.nb; def4 = make_float4(d.defv, d.defv, d.defv, d.defv);
        }
        const bool is_long = BPW == 1 || end - begin >= kPoolLong;
        float4 acc[C];
        float4 row[U][C];
        // ---- short bags: one tile per bag ----
        if constexpr (BPW == 4) {
            uint64_t i = is_long ? end : begin;
            bool first = true;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            // a short bag has fewer than 16 keys: its tile fetches them all with one coalesced load, lane tl holds key begin + tl
            const int64_t kpre = (!is_long && begin + tl < end) ? keys[begin + tl] : kEmpty;
            while (__any(i < end)) {  // wave-uniform; tiles whose bag is done idle through the ballots
                uint64_t pos[U];
                int64_t kv[U];
                bool inb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    pos[u] = i + u; inb[u] = pos[u] < end;
                    kv[u] = __shfl(kpre, tile * 16 + (int)((pos[u] - begin) & 15));
                }
                pooled_fetch<DIM4, U, C>(tkeys, values, nb, dim4, kv, pos, inb, tile, tl, def4, row, found, located, member);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!inb[u]) continue;
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (DIM4 != 0 || (uint32_t)(c * 16 + tl) < dim4) {
                            if (first) acc[c] = row[u][c];
                            else { acc[c].x += row[u][c].x; acc[c].y += row[u][c].y; acc[c].z += row[u][c].z; acc[c].w += row[u][c].w; }
                        }
                    first = false;
                }
                i += U;
            }
            if (has && !is_long) {
                const float len = (float)(end - begin);
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (DIM4 != 0 || (uint32_t)(c * 16 + tl) < dim4) {
                        float4 v = acc[c];
                        if (mean && end > begin) { v.x = v.x / len; v.y = v.y / len; v.z = v.z / len; v.w = v.w / len; }
                        out[bag * dim4 + c * 16 + tl] = v;
                    }
            }
        }
        // ---- long bags: the four tiles share one bag at a time ----
        const uint64_t long_mask = __ballot(is_long && has);
        if (!long_mask) continue;   // wave-uniform
        for (int q = 0; q < BPW; ++q) {
            if (!((long_mask >> (q * 16)) & 1)) continue;   // wave-uniform
            const uint64_t bq = __shfl(begin, q * 16), eq = __shfl(end, q * 16);
            if constexpr (GROUPED && BPW == 4) {   // all four tiles work for bag q's table now
                member = (b0 + q) / bags_per_table;
                const GroupDesc d = desc[member];
                tkeys = d.tkeys; values = d.values; nb = d.nb; def4 = make_float4(d.defv, d.defv, d.defv, d.defv);
            }
            bool first = true;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint64_t i = bq; i < eq; i += 4 * U) {   // wave-uniform
                uint64_t pos[U];
                int64_t kv[U];
                bool inb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { pos[u] = i + (uint64_t)u * 4 + tile; inb[u] = pos[u] < eq; kv[u] = inb[u] ? keys[pos[u]] : kEmpty; }
                pooled_fetch<DIM4, U, C>(tkeys, values, nb, dim4, kv, pos, inb, tile, tl, def4, row, found, located, member);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int src = 0; src < 4; ++src) {
                        if (i + (uint64_t)u * 4 + src >= eq) continue;   // wave-uniform
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            float4 v;   // every tile reads the row tile `src` fetched: all four keep the same running sum
                            v.x = __shfl(row[u][c].x, src * 16 + tl); v.y = __shfl(row[u][c].y, src * 16 + tl);
                            v.z = __shfl(row[u][c].z, src * 16 + tl); v.w = __shfl(row[u][c].w, src * 16 + tl);
                            if (first) acc[c] = v;
                            else { acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w; }
                        }
                        first = false;
                    }
            }
            if (tile == 0) {
                const float len = (float)(eq - bq);
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (DIM4 != 0 || (uint32_t)(c * 16 + tl) < dim4) {
                        float4 v = acc[c];
                        if (mean && eq > bq) { v.x = v.x / len; v.y = v.y / len; v.z = v.z / len; v.w = v.w / len; }
                        out[(b0 + q) * dim4 + c * 16 + tl] = v;
                    }
            }
        }
    }
}

}  // namespace mee

using namespace mee;

namespace mee {
// The library's own store policy for a dense output (nobody gave a hint).  One call cannot tell whether its output will be re-read from cache; the last few calls
// can: a ring of the latest calls' output buffers (base address, bytes).  The same buffer as the call before: the caller reuses one result buffer — cached stores
// while it fits (<= 128 MB: the Infinity Cache absorbs it).  Another buffer, and the distinct buffers of the last eight calls add up to more than 64 MB: the results
// rotate (a server's independent requests, a trainer's per-step activations), every one of them has to reach HBM and nothing re-reads it from cache — streaming
// stores (measured into 6 x 64 MB buffers: 38.6 -> 33.2 us per 256K-key find; two alternating 64 MB buffers: 35.4 cached).  mee_find_ex / "find_nt" stay authoritative.
static bool outputs_rotate(const mee_table* t, const void* d_out, uint64_t bytes) {
    mee_table* m = const_cast<mee_table*>(t);   // (policy state only; see meepo_table_int.h)
    const uint64_t p = (uint64_t)(uintptr_t)d_out;
    const uint32_t head = __atomic_load_n(&m->out_ring_head, __ATOMIC_RELAXED);
    if (__atomic_load_n(&m->out_ring_ptr[(head + 7u) & 7u], __ATOMIC_RELAXED) == p) return false;
    const uint32_t slot = __atomic_fetch_add(&m->out_ring_head, 1u, __ATOMIC_RELAXED) & 7u;
    __atomic_store_n(&m->out_ring_ptr[slot], p, __ATOMIC_RELAXED);
    __atomic_store_n(&m->out_ring_bytes[slot], bytes, __ATOMIC_RELAXED);
    uint64_t seen[8], total = 0;
    int distinct = 0;
    for (int i = 0; i < 8; ++i) {
        const uint64_t q = __atomic_load_n(&m->out_ring_ptr[i], __ATOMIC_RELAXED);
        if (!q) continue;
        bool dup = false;
        for (int j = 0; j < distinct; ++j) dup = dup || seen[j] == q;
        if (dup) continue;
        seen[distinct++] = q;
        total += __atomic_load_n(&m->out_ring_bytes[i], __ATOMIC_RELAXED);
    }
    return distinct >= 2 && total > (64ull << 20);
}

// every dense lookup of one plane (declared with its defaults in meepo_table_int.h: find_or_insert's first pass calls it too)
int find_plane(const mee_table* t, const float* plane, float miss_value, const int64_t* d_keys, size_t n, float* d_out,
               uint8_t* d_found, void* stream, bool missing_only, bool counted, bool rows_only,
               int64_t* d_slots_out, bool unordered, bool skip_padding, int nt_call /* this call's cache policy (mee_find_ex); -1: the table's */) {
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    int R = t->find_rounds > 0 ? t->find_rounds : (t->dim4 == 16 ? 2 : 1);
    if (t->dim4 != 16 && R > 4) R = 4;
    if (t->dim4 != 16 && t->dim4 != 32 && R > 2) R = 2;
    R = R >= 8 ? 8 : R >= 4 ? 4 : R >= 2 ? 2 : 1;
    const unsigned fblock = t->find_block == 64 || t->find_block == 128 ? (unsigned)t->find_block : 256u;   // (launch bound of find_kernel: 256)
    const unsigned grid = grid_for(n, (fblock / 64u) * 4u * (unsigned)R, t->find_grid_cap > 0 ? (unsigned)t->find_grid_cap : (1u << 22));
#define FIND1(D4, RR, NT) do { if (unordered) hipExtLaunchKernelGGL((find_kernel<D4, RR, NT>), dim3(grid), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, \
                                        (const int64_t*)t->keys, (const f32x4*)plane, t->nb, d_keys, (uint64_t)n, (f32x4*)d_out, d_found, miss_value, t->dim4, (uint32_t*)nullptr, (int64_t*)nullptr, (int64_t)0); \
                               else find_kernel<D4, RR, NT><<<grid, fblock, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, nullptr); } while (0)
    const int nt = nt_call >= 0 ? nt_call : t->find_nt >= 0 ? (t->find_nt & 7)
                 : ((uint64_t)n * t->dim * 4 <= (128ull << 20) && !outputs_rotate(t, d_out, (uint64_t)n * t->dim * 4) ? 4 : 0);
#define FIND(D4, RR) do { switch (nt) { case 0: FIND1(D4, RR, 0); break; case 1: FIND1(D4, RR, 1); break; case 2: FIND1(D4, RR, 2); break; case 3: FIND1(D4, RR, 3); break; case 4: FIND1(D4, RR, 4); break; case 5: FIND1(D4, RR, 5); break; case 6: FIND1(D4, RR, 6); break; default: FIND1(D4, RR, 7); } } while (0)
    if (skip_padding) {   // owner pass of a padded sharded exchange: EMPTY positions get neither a row nor a found byte (nobody reads them)
        const bool cached = nt & 4;
#define FINDP(D4, RR) do { if (cached) find_kernel<D4, RR, 132><<<grid, 256, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, nullptr); \
                           else find_kernel<D4, RR, 128><<<grid, 256, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, nullptr); } while (0)
        if (t->dim4 == 16) { if (R >= 2) FINDP(16, 2); else FINDP(16, 1); }
        else if (t->dim4 == 32) { if (R >= 2) FINDP(32, 2); else FINDP(32, 1); }
        else { if (R >= 2) FINDP(0, 2); else FINDP(0, 1); }
#undef FINDP
    } else
    if (d_slots_out) {   // located find: the plain kernel + one 8-byte store per key
        // cache policy of `out`: this is the forward of a TRAINING step — the apply that follows sweeps the Infinity Cache before the next
        // forward, so keeping the dense output cached buys nothing and streaming stores win (136.9 -> 132.5 us per find + Adagrad step)
        const bool cached_out = t->find_nt >= 0 && (t->find_nt & 4);
#define FINDL(D4, RR) do { if (cached_out) find_kernel<D4, RR, 68><<<grid, fblock, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, nullptr, d_slots_out, handle_tag_of(t)); \
                           else find_kernel<D4, RR, 64><<<grid, fblock, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, nullptr, d_slots_out, handle_tag_of(t)); } while (0)
        if (t->dim4 == 16) { if (R >= 2) FINDL(16, 2); else FINDL(16, 1); }
        else if (t->dim4 == 32) { if (R >= 2) FINDL(32, 2); else FINDL(32, 1); }
        else { if (R >= 2) FINDL(0, 2); else FINDL(0, 1); }
#undef FINDL
    } else
    if (missing_only) {
        const unsigned gm = grid_for(n, 256, 8192);
#define FMISS(F) find_missing_kernel<F><<<gm, 256, 0, st>>>(t->keys, (const float4*)plane, t->nb, t->dim4, d_keys, n, (float4*)d_out, d_found, t->hits)
        if (rows_only) FMISS(0); else if (counted) FMISS(3); else FMISS(1);
#undef FMISS
    } else if (counted) {  // sampled statistics pass: one key in flight per tile
        const unsigned g1 = grid_for(n, 16, 1u << 22);
#define FINDX(D4, NT) find_kernel<D4, 1, NT><<<g1, 256, 0, st>>>(t->keys, (const f32x4*)plane, t->nb, d_keys, n, (f32x4*)d_out, d_found, miss_value, t->dim4, t->hits)
#define FINDX_D(NT) do { if (t->dim4 == 16) FINDX(16, NT); else if (t->dim4 == 32) FINDX(32, NT); else FINDX(0, NT); } while (0)
        FINDX_D(20);
#undef FINDX_D
#undef FINDX
    } else
    if (t->dim4 == 16) { if (R == 8) FIND(16, 8); else if (R == 4) FIND(16, 4); else if (R == 2) FIND(16, 2); else FIND(16, 1); }
    else if (t->dim4 == 32) { if (R == 4) FIND(32, 4); else if (R == 2) FIND(32, 2); else FIND(32, 1); }
    else { if (R == 2) FIND(0, 2); else FIND(0, 1); }
#undef FIND
#undef FIND1
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

}  // namespace mee
extern "C" {

int mee_find(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find");
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find: null argument");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream);
}

int mee_find_ex(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, uint32_t flags, void* stream) {
    MEE_RANGE("mee_find_ex");
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_ex: null argument");
    if (flags & ~(uint32_t)(MEE_FIND_STREAM_STORES | MEE_FIND_CACHED_STORES | MEE_FIND_STREAM_ROWS | MEE_FIND_STREAM_BUCKETS))
        return fail(MEE_ERR_INVALID_ARG, "mee_find_ex: unknown flag bits 0x%x", flags);
    if ((flags & MEE_FIND_STREAM_STORES) && (flags & MEE_FIND_CACHED_STORES))
        return fail(MEE_ERR_INVALID_ARG, "mee_find_ex: MEE_FIND_STREAM_STORES and MEE_FIND_CACHED_STORES exclude each other");
    if (flags == MEE_FIND_DEFAULT) return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream);
    // the kernel's policy bits: 1 = streaming row loads, 2 = streaming bucket loads, 4 = cached stores of the dense output
    const bool cached_out = (flags & MEE_FIND_CACHED_STORES) ||
                            (!(flags & MEE_FIND_STREAM_STORES) && (uint64_t)n * t->dim * 4 <= (128ull << 20) && !outputs_rotate(t, d_out, (uint64_t)n * t->dim * 4));
    const int nt = (flags & MEE_FIND_STREAM_ROWS ? 1 : 0) | (flags & MEE_FIND_STREAM_BUCKETS ? 2 : 0) | (cached_out ? 4 : 0);
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, false, false, false, nullptr, false, false, nt);
}

}  // extern "C"
namespace mee {
// mee_find for the owner side of a padded sharded exchange (meepo_sharded.hip): MEE_EMPTY_KEY positions are padding that nobody reads —
// they get neither a default row nor a found byte
int find_skip_padding(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    if (!t || (n && (!d_keys || !d_out || !d_found))) return fail(MEE_ERR_INVALID_ARG, "find_skip_padding: null argument");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, false, false, false, nullptr, false, /*skip_padding=*/true);
}
}  // namespace mee
extern "C" {

int mee_find_located(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream) {
    MEE_RANGE("mee_find_located");
    if (!t || (n && (!d_keys || !d_out || !d_slots_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_located: null argument");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, false, false, false, d_slots_out);
}

int mee_find_many(const mee_table* t, const mee_find_request* reqs, uint32_t count, void* stream) {
    MEE_RANGE("mee_find_many");
    if (!t || !reqs) return fail(MEE_ERR_INVALID_ARG, "mee_find_many: null argument");
    if (count == 0) return MEE_OK;
    if (count > (uint32_t)kMaxFindRequests) return fail(MEE_ERR_INVALID_ARG, "mee_find_many: %u requests (at most %d per call)", count, kMaxFindRequests);
    FindMany m{};
    const int R = t->dim4 == 16 ? 2 : 1;
    uint64_t blocks = 0, out_bytes = 0;
    uint32_t used = 0;
    for (uint32_t q = 0; q < count; ++q) {
        if (reqs[q].n == 0) continue;
        if (!reqs[q].d_keys || !reqs[q].d_out) return fail(MEE_ERR_INVALID_ARG, "mee_find_many: request %u has a null buffer", q);
        m.keys[used] = reqs[q].d_keys; m.out[used] = (f32x4*)reqs[q].d_out; m.found[used] = reqs[q].d_found; m.n[used] = reqs[q].n;
        m.first_block[used] = (uint32_t)blocks;
        blocks += (reqs[q].n + 16ull * R - 1) / (16ull * R);
        out_bytes += (uint64_t)reqs[q].n * t->dim * 4;
        ++used;
    }
    if (used == 0) return MEE_OK;
    if (blocks > (1ull << 31)) return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_find_many: %llu blocks", (unsigned long long)blocks);
    m.first_block[used] = (uint32_t)blocks; m.count = used;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    const bool cached_out = out_bytes <= (128ull << 20);   // same policy as mee_find, on the total output of the launch
#define FM(D4, RR) do { if (cached_out) find_many_kernel<D4, RR, 4><<<(unsigned)blocks, 256, 0, st>>>(t->keys, (const f32x4*)t->values, t->nb, m, t->default_value, t->dim4); \
                        else find_many_kernel<D4, RR, 0><<<(unsigned)blocks, 256, 0, st>>>(t->keys, (const f32x4*)t->values, t->nb, m, t->default_value, t->dim4); } while (0)
    if (t->dim4 == 16) FM(16, 2); else if (t->dim4 == 32) FM(32, 1); else FM(0, 1);
#undef FM
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_find_unordered(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find_unordered");
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_unordered: null argument");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, false, false, false, nullptr, /*unordered=*/true);
}

int mee_find_missing(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find_missing");
    if (!t || (n && (!d_keys || !d_out || !d_found))) return fail(MEE_ERR_INVALID_ARG, "mee_find_missing: null argument");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, true);
}

int mee_find_counted(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int missing_only, void* stream) {
    MEE_RANGE("mee_find_counted");
    if (!t || (n && (!d_keys || !d_out || !d_found))) return fail(MEE_ERR_INVALID_ARG, "mee_find_counted: null argument");
    if (!t->hits) return fail(MEE_ERR_UNSUPPORTED, "mee_find_counted: table was created without MEE_FLAG_TRACK_HITS");
    return find_plane(t, t->values, t->default_value, d_keys, n, d_out, d_found, stream, missing_only != 0, true);
}

int mee_find_pooled(const mee_table* t, const int64_t* d_keys, size_t n, const uint64_t* d_bag_offsets, size_t n_bags, float* d_out,
                    uint8_t* d_found, int mode, void* stream) {
    MEE_RANGE("mee_find_pooled");
    if (!t || (n_bags && (!d_bag_offsets || !d_out)) || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_find_pooled: null argument");
    if (mode != MEE_POOL_SUM && mode != MEE_POOL_MEAN) return fail(MEE_ERR_INVALID_ARG, "mee_find_pooled: mode must be MEE_POOL_SUM or MEE_POOL_MEAN");
    if (n_bags == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    // n (the number of keys, a host value) only picks the launch shape: mostly short bags -> four bags per wave, long average -> one
    const bool wave_per_bag = n / n_bags >= 12;
#define POOLED(D4, U1, U4) do { if (wave_per_bag) find_pooled_kernel<D4, U1, 1><<<grid_for(n_bags, 4, 1u << 20), 256, 0, st>>>(t->keys, (const float4*)t->values, t->nb, d_keys, d_bag_offsets, n_bags, (float4*)d_out, d_found, t->default_value, t->dim4, mode == MEE_POOL_MEAN, nullptr, 1, nullptr, n); \
                                else find_pooled_kernel<D4, U4, 4><<<grid_for(n_bags, 16, 1u << 20), 256, 0, st>>>(t->keys, (const float4*)t->values, t->nb, d_keys, d_bag_offsets, n_bags, (float4*)d_out, d_found, t->default_value, t->dim4, mode == MEE_POOL_MEAN, nullptr, 1, nullptr, n); } while (0)
    if (t->dim4 == 16) POOLED(16, 4, 2); else if (t->dim4 == 32) POOLED(32, 2, 1); else POOLED(0, 1, 1);
#undef POOLED
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int mee_find_plane(const mee_table* t, uint32_t plane, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream) {
    MEE_RANGE("mee_find_plane");
    if (!t || (n && (!d_keys || !d_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_plane: null argument");
    const float* p = plane_of(t, plane);
    if (!p) return fail(MEE_ERR_UNSUPPORTED, "mee_find_plane: plane %u does not exist (optimizer=%u)", plane, t->optimizer);
    return find_plane(t, p, plane == 0 ? t->default_value : 0.0f, d_keys, n, d_out, d_found, stream);
}

// ---- the embedding-bag collection: pooled lookups of a whole group in one launch, and their backward -----------------------
int mee_group_find_pooled(mee_group* g, const int64_t* d_keys, size_t n, const uint64_t* d_bag_offsets, size_t bags_per_table,
                          float* d_out, uint8_t* d_found, int64_t* d_located_out, int mode, void* stream) {
    MEE_RANGE("mee_group_find_pooled");
    if (!g || (bags_per_table && (!d_bag_offsets || !d_out)) || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_group_find_pooled: null argument");
    if (mode != MEE_POOL_SUM && mode != MEE_POOL_MEAN) return fail(MEE_ERR_INVALID_ARG, "mee_group_find_pooled: mode must be MEE_POOL_SUM or MEE_POOL_MEAN");
    if (bags_per_table == 0) return MEE_OK;
    if (int rc = group_refresh(g, stream)) return rc;
    DeviceGuard guard(g->device);
    hipStream_t st = as_stream(stream);
    const uint64_t n_bags = (uint64_t)g->n_tables * bags_per_table;
    const bool wave_per_bag = n / n_bags >= 12;
#define GPOOLED(D4, U1, U4) do { if (wave_per_bag) find_pooled_kernel<D4, U1, 1, true><<<grid_for(n_bags, 4, 1u << 20), 256, 0, st>>>(nullptr, nullptr, 0, d_keys, d_bag_offsets, n_bags, (float4*)d_out, d_found, 0.f, g->dim4, mode == MEE_POOL_MEAN, g->d_desc, bags_per_table, d_located_out, n); \
                                 else find_pooled_kernel<D4, U4, 4, true><<<grid_for(n_bags, 16, 1u << 20), 256, 0, st>>>(nullptr, nullptr, 0, d_keys, d_bag_offsets, n_bags, (float4*)d_out, d_found, 0.f, g->dim4, mode == MEE_POOL_MEAN, g->d_desc, bags_per_table, d_located_out, n); } while (0)
    if (g->dim4 == 16) GPOOLED(16, 4, 2); else if (g->dim4 == 32) GPOOLED(32, 2, 1); else GPOOLED(0, 1, 1);
#undef GPOOLED
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// The training forward: mee_find_located whose launch also carries mee_apply_prepare for the SAME keys (the partition half of the bucketed
// apply, run by the launch's first blocks beside the row gather).  (Table without optimizer: plain mee_find_located.)
int mee_find_located_prepare(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream) {
    MEE_RANGE("mee_find_located_prepare");
    if (!t || (n && (!d_keys || !d_out || !d_slots_out))) return fail(MEE_ERR_INVALID_ARG, "mee_find_located_prepare: null argument");
    if (t->prepared_n) return fail(MEE_ERR_INVALID_ARG, "mee_find_located_prepare: a prepared apply is already pending");
    if (n == 0) return MEE_OK;
    if (t->optimizer == MEE_OPT_NONE) return mee_find_located(t, d_keys, n, d_out, d_found, d_slots_out, stream);
    if (n > t->max_batch) return fail(MEE_ERR_BATCH_TOO_LARGE, "mee_find_located_prepare: n=%zu exceeds config.max_batch=%llu", n, (unsigned long long)t->max_batch);
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    uint32_t apply_grid, nbk;
    bool apply_full;
    const uint32_t nbk_hash = bucket_count_for(t, n, st, &apply_grid, &nbk, &apply_full);
    uint32_t part_blocks, per_block;
    part_geometry((uint32_t)n, kFindPrepareThreads, part_blocks, per_block);
    const int R = t->dim4 == 16 ? kFindPrepareR : t->dim4 == 32 ? 2 : 1;
    const unsigned find_cap = t->prepare_debug >> 8;
    const unsigned find_blocks = grid_for(n, (kFindPrepareThreads / 64) * 4u * (unsigned)R, find_cap ? find_cap : 1u << 22);
    const bool separate = t->prepare_debug & 1;
    if (separate) part_blocks = 0;
    const bool cached_out = t->find_nt >= 0 && (t->find_nt & 4);
#define FINDLP1(D4, RR, NT) find_prepare_kernel<D4, RR, NT><<<part_blocks + find_blocks, kFindPrepareThreads, sizeof(PartHot) + nbk * 4, st>>>(t->keys, (const f32x4*)t->values, t->nb, d_keys, n, (f32x4*)d_out, d_found, \
        t->default_value, t->dim4, d_slots_out, handle_tag_of(t), part_blocks, nbk_hash, nbk, per_block, t->bk, &t->ctr->status, t->op, t->bk.xcd_split)
#define FINDLP(D4, RR) do { if (cached_out) FINDLP1(D4, RR, 68); else FINDLP1(D4, RR, 64); } while (0)
    if (t->dim4 == 16) FINDLP(16, kFindPrepareR); else if (t->dim4 == 32) FINDLP(32, 2); else FINDLP(0, 1);
#undef FINDLP
#undef FINDLP1
    MEE_HIP(hipGetLastError());
    if (separate) { if (int rc = bucket_apply_prepare(t, d_keys, (uint32_t)n, st)) return rc; }
    else { t->part_blocks = part_blocks; t->part_per_block = per_block; t->part_nbk = nbk; t->part_nbk_hash = nbk_hash; t->part_grid = apply_grid; t->part_full = apply_full; }
    t->prepared_n = n; t->prepared_keys = d_keys; t->prepared_by_forward = true;
    return MEE_OK;
}

#if MEE_FIND_TIMELINE
int mee_debug_find_timeline(unsigned long long* host_out, uint64_t n_words) {   // first call arms the buffer, later calls read it
    static unsigned long long* buf = nullptr;
    if (!buf) {
        if (hipMalloc((void**)&buf, 16384 * 4 * 8) != hipSuccess) return 1;
        (void)hipMemset(buf, 0, 16384 * 4 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_find_dbg), &buf, sizeof buf);
        return 0;
    }
    return hipMemcpy(host_out, buf, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

}  // extern "C"

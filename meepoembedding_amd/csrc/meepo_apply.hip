// meepo_apply.hip — the bucketed sparse-optimizer apply (SPEC.md §4): duplicate-key reduction without a global group table.
//
// Reference anchor: /root/reference/README.md:2 (no code upstream); BASELINE.json north_star: "the sparse-optimizer (Adagrad/Adam)
// scatter-update … wavefront ballot/prefix-sum for duplicate-key reduction".
//
// The group-table apply (meepo_table.hip) pays, per key of a batch, one scattered atomic to claim an entry of a global table, one scattered
// read of that entry in the main pass and one scattered store to release it — ~30 us of a 256K-key batch that has (almost) no duplicates —
// plus three dependent launches for the duplicates.  Here the batch is first PARTITIONED by the top bits of mix64(key) into buckets of
// 128..352 positions (one small kernel, LDS histograms, no per-key global atomic), and then ONE kernel gives every bucket to one 512-thread
// block: all occurrences of a key are in the same bucket, so the block finds the batch's duplicates in an LDS hash table, sorts the
// bucket's positions by key with LDS prefix sums, and its 32 tiles update each distinct key once — a key that occurs once straight from
// its gradient row (bit-exact), a key that occurs c times from the fp64 sum of its c rows (runs of more than 32 are cut into chunks
// whose fp64 partial rows the block combines itself).  Two launches per apply whatever the key distribution (one behind the training
// forward, whose launch carries the partition), no per-key global atomic, nothing to clean up afterwards.
//
// A bucket that a hot key makes larger than one block's LDS (512 positions) is SPLIT: its slabs of 512 positions go to blocks of their
// own (the spare blocks at the head of the grid, so they start first), each slab emits one pending record per distinct key (key + fp64 partial sum), and the slab
// that finishes last — an agent-scope release / ticket / acquire hand-off, no block ever waits for another — merges the bucket's records
// with the same LDS machinery and applies the updates.  A key with 21 000 occurrences in a 256K-key batch (Zipf 1.05) is summed by 41
// blocks on 41 CUs, not by one.
//
// Bucket = top bits of mix64(key) = the bits that pick the key's table bucket (mulhi64): a block's keys live in one contiguous 1/n_buckets
// slice of the table.
#include <hip/hip_runtime.h>

#include "meepo_apply_part.h"

namespace mee {

#ifndef MEE_PT
#define MEE_PT 1024
#endif
constexpr int kPartThreads = MEE_PT;
constexpr int kApplyThreads = 512;
constexpr int kApplyWaves = kApplyThreads / 64;
constexpr uint32_t kLdsSlots = 2 * kSlab;
constexpr uint32_t kRun = 32;             // occurrences one tile sums in one go; longer runs are cut into chunks of this many

uint32_t bucket_count_for(const mee_table* t, uint64_t n) { return bucket_count_for_host(n, t->bk.slots, t->bk.bucket_max ? t->bk.bucket_max : kBucketMax); }
static uint32_t max_extra_slabs(uint64_t n) { return 2 * (uint32_t)(n / kSlab) + 2; }   // slabs of split buckets: sum of ceil(size / kSlab) over buckets larger than kSlab

// ---- the partition kernel of an apply (the role itself: meepo_apply_part.h) ----------------------------------------------------------------
__global__ __launch_bounds__(kPartThreads) void bkt_sort_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t nbk,
                                                                uint32_t per_block, BucketScratch bk, uint32_t* status, OpCounters* op) {
    extern __shared__ uint32_t cursor[];
    __shared__ unsigned long long wsum[kPartThreads / 64];
    sort_role<kPartThreads>(keys, n, nbk, per_block, blockIdx.x, gridDim.x, bk, status, op, cursor, wsum);
}

// ---- the apply kernel -------------------------------------------------------------------------------------------------------------
struct ApplyLds {
    unsigned long long key[kLdsSlots];   // key ^ kBias, 0 = empty
    long long slot[kLdsSlots];           // per run: the table slot its handle names (LOCATED kernels)
    uint32_t cnt[kLdsSlots];             // occurrences of the run's key in this slab
    uint32_t off[kLdsSlots];             // where the run starts in src
    uint32_t run[kLdsSlots];             // slabs of split buckets: the run's pending-record number inside the slab
    uint32_t src[kSlab];                 // the slab's sources sorted by run: gradient-row index (positions) | pending-record index (merge)
    uint32_t items[kSlab + 32];          // work items: run slot | chunk << 10 | partial row << 15
    uint32_t big[32];                    // runs longer than kRun: run slot | first partial row << 10 | chunks << 20
    unsigned long long wsum[kApplyWaves];
    unsigned long long stk_val[72];      // merge: hash prefixes still to do
    uint32_t stk_bits[72];
    unsigned long long kmin, kmax;       // merge: smallest / largest biased key among a pass's candidates (equal: the pass holds ONE key)
    uint32_t seg_first[kPartBlocks + 1]; // where partition block k's entries of this bucket begin within the bucket (prefix of the run lengths)
    uint32_t seg_at[kPartBlocks];        // ... and where that run lies in pos / pkey
    uint32_t n_items, n_big, n_runs, part_base, rec_base, is_last, n_cand, stk_n;
};

// MEE_APPLY_TIMELINE (diagnostic builds only: tools/apply_timeline.py): thread 0 of every block stamps the 100 MHz wall clock at the phase
// boundaries of its bucket into a buffer the host reads back (mee_debug_timeline).
#ifndef MEE_APPLY_TIMELINE
#define MEE_APPLY_TIMELINE 0
#endif
#if MEE_APPLY_TIMELINE
#define MEE_TL(A_, i) do { if (threadIdx.x == 0) (A_).dbg[(uint64_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MEE_TL(A_, i) do { } while (0)
#endif

// MEE_APPLY_STORE_MODE (diagnostic builds): how an update's rows are stored — 0 plain, 1 nt, 2 sc1 (write-through, the line leaves the XCD's L2)
#ifndef MEE_APPLY_STORE_MODE
#define MEE_APPLY_STORE_MODE 0
#endif
__device__ __forceinline__ void store_row4(float4* p, const float4& v) {
#if MEE_APPLY_STORE_MODE == 1
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
#elif MEE_APPLY_STORE_MODE == 2
    const f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
#else
    *p = v;
#endif
}

struct ApplyArgs {
    const int64_t* tkeys; float4 *values, *s1, *s2; uint64_t nb; uint32_t dim4;
    const float4* grads; const uint32_t* gidx; const int64_t* slots;
    uint64_t capacity; int64_t handle_tag; uint32_t* status;
    double* part; uint32_t max_part;     // fp64 partial rows of long runs (BatchScratch::gacc)
    uint32_t nbk, part_blocks, per_block;   // the partition: buckets, partition blocks, batch positions per partition block
    uint32_t n_extra;                       // spare blocks at the head of the grid
    OpCounters* op;
#if MEE_APPLY_TIMELINE
    unsigned long long* dbg;
#endif
    const GroupDesc* desc; uint32_t n_tables;   // GROUPED kernels (mee_group_apply_*): the members' planes; a "slot" is member << 48 | slot
    OptArgs a;
};

// Where a finished run's row lives.  Plain tables: the table's planes, `slot` as it is.  GROUPED (the apply of a table group: the batch "keys"
// are located rows already, member << 48 | slot, and serve as slot handles too): the member's planes — its descriptor comes out of LDS when the
// group is small enough to be staged there (gdesc != nullptr), else from device memory — and the low 48 bits.
struct RowAt { float4 *values, *s1, *s2; uint64_t row; };
template <bool GROUPED>
__device__ __forceinline__ RowAt row_at(const ApplyArgs& A, const GroupDesc* gdesc, int64_t slot, bool upd) {
    RowAt r{A.values, A.s1, A.s2, upd ? (uint64_t)slot : 0ull};
    if constexpr (GROUPED) {
        const uint64_t member = upd ? (uint64_t)slot >> kGroupSlotBits : 0ull;
        const GroupDesc& d = gdesc ? gdesc[member] : A.desc[member];
        r.values = d.values; r.s1 = d.s1; r.s2 = d.s2;
        r.row = upd ? (uint64_t)slot & ((1ull << kGroupSlotBits) - 1) : 0ull;
    }
    return r;
}

// One slab: m <= kSlab sources -> one update (or one pending record) per distinct key.
//   src_rec = false: the sources are entries [first, first + m) of bucket b (batch positions + keys, pulled out of the partition blocks' slices
//                    of bk.pos / bk.pkey); a source's row is a row of grads.
//   src_rec = true : (merge of a split bucket) the sources are the pending records whose bucket-relative numbers lie in L.src[0 .. m); a
//                    source's row is the record's fp64 partial sum.
//   emit: results become pending records of bucket `b` instead of table updates (slab of a split bucket).
// ONE inlined instance per kernel (the kernel loops over it: slab first, merge passes after): as two instances the merge copy pushed the
// kernel from 64 to 113 VGPRs, i.e. the hot path of every block from 4 to 2 resident blocks per CU (-10 us per 256K-key batch).
template <int KIND, int DIM4, bool LOCATED, bool SPLIT /* false: emit and src_rec are known to be false (the kernel of the whole buckets) */, bool GROUPED = false>
__device__ __forceinline__ void process_slab(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, uint32_t first, uint32_t m, bool emit_rt, bool src_rec_rt,
                                             uint32_t b, uint32_t rec_bucket0 /* split buckets: the bucket's first pending record */, const GroupDesc* gdesc = nullptr) {
    static_assert(!GROUPED || LOCATED, "a group's batch names rows, not keys");
    const bool emit = SPLIT && emit_rt, src_rec = SPLIT && src_rec_rt;
    // The kernel calls this from a loop (slab, then merge passes).  The thread index is re-read through an empty asm in every call so that
    // nothing derived from it looks loop-invariant: hoisted out of that loop, the per-thread address arithmetic of every array touched in
    // here stayed live across the whole kernel (110 VGPRs instead of 64: half the resident blocks per CU for every block's hot path).
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, tile = lane >> 4, tl = lane & 15, wv = t >> 6;
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4;
    OptArgs a = A.a;
    a.kind = KIND;
    // ---- 1. the slab's keys into the LDS hash table: run = LDS slot, r = arrival number inside the run ----
    uint32_t my_src = 0;
    if (src_rec) my_src = t < m ? L.src[t] : 0u;   // read before the table is cleared / src is rewritten
    __syncthreads();   // (src_rec = false: the caller's wave 0 has just written L.seg_first / L.seg_at)
    for (uint32_t j = t; j < kLdsSlots; j += kApplyThreads) { L.key[j] = 0ull; L.cnt[j] = 0u; }
    if (t == 0) { L.n_big = 0u; }
    __syncthreads();
    uint32_t my_slot = 0, my_r = 0;
    bool my_first = false;
    int64_t my_tslot = -1;
    if (t < m) {
        int64_t key;
        int64_t tslot = -1;   // LOCATED: the key's slot handle (positions) | the table slot its record carries (merge)
        if (src_rec) {
            key = bk.pend_key[rec_bucket0 + my_src];
            if constexpr (LOCATED) tslot = bk.pend_slot[rec_bucket0 + my_src];
        } else {
            const uint32_t gi = first + t;   // index within the bucket -> its run (binary search over the 64 run starts) -> its place in pos / pkey
            uint32_t k = 0;
#pragma unroll
            for (uint32_t stp = kPartBlocks / 2; stp; stp >>= 1) if (L.seg_first[k + stp] <= gi) k += stp;
            const uint32_t at = L.seg_at[k] + (gi - L.seg_first[k]);
            const uint32_t p = bk.pos[at];
            key = bk.pkey[at];
            my_src = A.gidx ? min(A.gidx[p], a.grad_rows - 1) : p;   // the row of the grad array that belongs to the position
            if constexpr (LOCATED) tslot = A.slots[p];   // the raw handle: decoded where it is first needed, so that the load travels beside the LDS work
        }
        const unsigned long long bkey = (unsigned long long)key ^ kBias;
        my_slot = (uint32_t)(mix64b((uint64_t)key) & (kLdsSlots - 1));   // mix64's top bits chose the bucket: take another mixer here
        bool first_of_run = false;
        while (true) {
            const unsigned long long old = atomicCAS(&L.key[my_slot], 0ull, bkey);
            if (old == 0ull) { first_of_run = true; break; }
            if (old == bkey) break;
            my_slot = (my_slot + 1) & (kLdsSlots - 1);
        }
        my_r = atomicAdd(&L.cnt[my_slot], 1u);
        my_first = first_of_run;
        my_tslot = tslot;
    }
    __syncthreads();
    if constexpr (!SPLIT) MEE_TL(A, 2);   // entries fetched, keys in the LDS table
    // ---- 2. prefix sums over the runs: where each run starts in src, its work items, its number, its fp64 partial rows ----
    {
        const uint32_t s0 = 2 * t, c0 = L.cnt[s0], c1 = L.cnt[s0 + 1];
        const uint32_t i0 = (c0 + kRun - 1) / kRun, i1 = (c1 + kRun - 1) / kRun;
        const uint32_t p0 = c0 > kRun ? i0 : 0u, p1 = c1 > kRun ? i1 : 0u;
        // bits 0..15 sources | 16..31 items | 32..47 runs | 48..63 partial rows — each at most kSlab + kSlab / kRun
        const unsigned long long packed = (unsigned long long)(c0 + c1) | (unsigned long long)(i0 + i1) << 16 |
                                          (unsigned long long)((c0 != 0) + (c1 != 0)) << 32 | (unsigned long long)(p0 + p1) << 48;
        unsigned long long total;
        const unsigned long long ex = block_scan_u64<kApplyWaves>(packed, L.wsum, total);
        uint32_t so = (uint32_t)ex & 0xFFFFu, io = (uint32_t)(ex >> 16) & 0xFFFFu, ro = (uint32_t)(ex >> 32) & 0xFFFFu, po = (uint32_t)(ex >> 48);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t s = s0 + h, c = h ? c1 : c0, ni = h ? i1 : i0;
            L.off[s] = so;
            if (c) {
                L.run[s] = ro;
#pragma unroll 1   // (unrolled 16-fold this loop alone held 40 VGPRs of precomputed item words)
                for (uint32_t j = 0; j < ni; ++j) L.items[io + j] = s | j << 10 | (c > kRun ? (po + j) << 15 : 0u);
                if (c > kRun) { L.big[atomicAdd(&L.n_big, 1u)] = s | po << 10 | ni << 20; po += ni; }
                so += c; io += ni; ++ro;
            }
        }
        if (t == 0) {
            L.n_items = (uint32_t)(total >> 16) & 0xFFFFu;
            L.n_runs = (uint32_t)(total >> 32) & 0xFFFFu;
            const uint32_t np = (uint32_t)(total >> 48);
            uint32_t pb = np ? atomicAdd(&A.op->n_part, np) : 0u;
            if (pb + np > A.max_part) pb = 0u;   // cannot happen (sum of ceil(c / kRun) over runs longer than kRun <= n / 32 + n / 33); never write out of bounds
            L.part_base = pb;
            if (emit) L.rec_base = atomicAdd(&bk.pend_cnt[b], L.n_runs);
        }
    }
    __syncthreads();
    if (t < m) {
        L.src[L.off[my_slot] + my_r] = my_src;
        if constexpr (LOCATED) {
            if (my_first) {   // every occurrence of a key names the same slot: the run keeps one
                if (src_rec) L.slot[my_slot] = my_tslot;
                else if constexpr (GROUPED) L.slot[my_slot] = my_tslot >= 0 && ((uint64_t)my_tslot >> kGroupSlotBits) < A.n_tables ? my_tslot : -1;
                else {
                    bool stale;
                    L.slot[my_slot] = handle_slot(my_tslot, A.handle_tag, A.capacity, stale);
                    if (stale) atomicOr(A.status, (uint32_t)MEE_STATUS_STALE_HANDLE);   // rare: the caller kept handles across a remove / clear / reserve
                }
            }
        }
    }
    __syncthreads();
    if constexpr (!SPLIT) MEE_TL(A, 3);   // scans done, sources sorted (LOCATED: slot handles arrived)
    const uint32_t n_items = L.n_items, n_big = L.n_big;
    const uint32_t rec_out0 = rec_bucket0 + (emit ? L.rec_base : 0u);   // emit: where this slab's records go
    // ---- 3. work items: one tile each ----
    for (uint32_t it0 = (uint32_t)wv * 4; it0 < n_items; it0 += kApplyWaves * 4) {   // block-uniform bound: the ballots inside tile_locate need whole waves
        const uint32_t item = it0 + tile;
        const bool valid = item < n_items;
        const uint32_t e = valid ? L.items[item] : 0u;
        const uint32_t s = e & 1023u, chunk = (e >> 10) & 31u, prow = e >> 15;
        const uint32_t c = valid ? L.cnt[s] : 0u;
        const uint32_t run0 = L.off[s] + kRun * chunk;
        const uint32_t nh = valid ? min(kRun, c - kRun * chunk) : 0u;
        const bool whole = valid && c <= kRun;   // the run is this one item
        const bool fin = whole && !emit;         // finished here: locate the row, update it once
        const bool single = !src_rec && fin && c == 1;
        const int64_t key = (int64_t)(L.key[s] ^ kBias);
        const uint32_t src0 = valid ? L.src[run0] : 0u;
        f32x4 gpre = {0.f, 0.f, 0.f, 0.f};
        // a key that occurs once (the bulk): its gradient row is read exactly once — stream it, and request it before anything else
        if (single && (uint32_t)tl < dim4) gpre = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A.grads) + (uint64_t)src0 * dim4 + tl);
        int64_t slot = -1;
        if constexpr (LOCATED) slot = valid ? (int64_t)L.slot[s] : -1;
        else {
            bool is_new, full;
            slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, fin, tile, tl, is_new, full);
        }
        if (!valid) continue;
        const bool upd = fin && slot >= 0;
        const RowAt at = row_at<GROUPED>(A, gdesc, slot, upd);
        for (uint32_t col = tl; col < dim4; col += 16) {
            const uint64_t o = at.row * dim4 + col;
            if (single) {   // gradient row (already requested) + the key's row -> update -> store; nothing else is live here
                if (upd) {
                    float4 w = at.values[o], x1 = at.s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
                    const float4 g = col == (uint32_t)tl ? make_float4(gpre.x, gpre.y, gpre.z, gpre.w) : A.grads[(uint64_t)src0 * dim4 + col];
                    opt_update4(a, w, x1, x2, g);
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                }
                continue;
            }
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            if (src_rec) {
                for (uint32_t q0 = 0; q0 < nh; q0 += 2) {   // two fp64 rows in flight; a lane past the end reads the last row again and adds +0.0
                    double2 lo[2], hi[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const double2* r = reinterpret_cast<const double2*>(bk.pend_row + ((uint64_t)(rec_bucket0 + L.src[run0 + min(q0 + q, nh - 1)]) * dim4 + col) * 4);
                        lo[q] = r[0]; hi[q] = r[1];
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bool on = q0 + q < nh;
                        sx += on ? lo[q].x : 0.0; sy += on ? lo[q].y : 0.0; sz += on ? hi[q].x : 0.0; sw += on ? hi[q].y : 0.0;
                    }
                }
            } else {
                for (uint32_t q0 = 0; q0 < nh; q0 += 4) {   // four rows in flight, no one-row-at-a-time tail (see chunk_sum in meepo_table.hip)
                    float4 gq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gq[q] = A.grads[(uint64_t)L.src[run0 + min(q0 + q, nh - 1)] * dim4 + col];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool on = q0 + q < nh;
                        sx += on ? (double)gq[q].x : 0.0; sy += on ? (double)gq[q].y : 0.0; sz += on ? (double)gq[q].z : 0.0; sw += on ? (double)gq[q].w : 0.0;
                    }
                }
            }
            if (fin) {
                if (upd) {
                    float4 w = at.values[o], x1 = at.s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
                    opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                }
            } else {   // a chunk of a long run -> one fp64 partial row of this block | a whole run of a split bucket's slab -> its pending record
                double2* dst = whole ? reinterpret_cast<double2*>(bk.pend_row + ((uint64_t)(rec_out0 + L.run[s]) * dim4 + col) * 4)
                                     : reinterpret_cast<double2*>(A.part + ((uint64_t)(L.part_base + prow) * dim4 + col) * 4);
                dst[0] = make_double2(sx, sy); dst[1] = make_double2(sz, sw);
            }
        }
        if (!fin && whole && tl == 0) {
            bk.pend_key[rec_out0 + L.run[s]] = key;
            if constexpr (LOCATED) bk.pend_slot[rec_out0 + L.run[s]] = slot;
        }
    }
    if constexpr (!SPLIT) MEE_TL(A, 4);   // this thread's items done
    if (n_big == 0) return;   // block-uniform
    // ---- 4. runs longer than kRun: one tile adds the run's chunk sums (written by this block: visible after the barrier) and finishes the run ----
    __syncthreads();
    for (uint32_t k0 = (uint32_t)wv * 4; k0 < n_big; k0 += kApplyWaves * 4) {
        const uint32_t k = k0 + tile;
        const bool valid = k < n_big;
        const uint32_t e = valid ? L.big[k] : 0u;
        const uint32_t s = e & 1023u, p0 = (e >> 10) & 1023u, np = e >> 20;
        const bool fin = valid && !emit;
        const int64_t key = (int64_t)(L.key[s] ^ kBias);
        int64_t slot = -1;
        if constexpr (LOCATED) slot = valid ? (int64_t)L.slot[s] : -1;
        else {
            bool is_new, full;
            slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, fin, tile, tl, is_new, full);
        }
        if (!valid) continue;
        const bool upd = fin && slot >= 0;
        const RowAt at = row_at<GROUPED>(A, gdesc, slot, upd);
        for (uint32_t col = tl; col < dim4; col += 16) {
            const uint64_t o = at.row * dim4 + col;
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            for (uint32_t q0 = 0; q0 < np; q0 += 2) {
                double2 lo[2], hi[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const double2* r = reinterpret_cast<const double2*>(A.part + ((uint64_t)(L.part_base + p0 + min(q0 + q, np - 1)) * dim4 + col) * 4);
                    lo[q] = r[0]; hi[q] = r[1];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool on = q0 + q < np;
                    sx += on ? lo[q].x : 0.0; sy += on ? lo[q].y : 0.0; sz += on ? hi[q].x : 0.0; sw += on ? hi[q].y : 0.0;
                }
            }
            if (fin) {
                if (upd) {
                    float4 w = at.values[o], x1 = at.s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
                    opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                }
            } else {
                double2* dst = reinterpret_cast<double2*>(bk.pend_row + ((uint64_t)(rec_out0 + L.run[s]) * dim4 + col) * 4);
                dst[0] = make_double2(sx, sy); dst[1] = make_double2(sz, sw);
            }
        }
        if (!fin && tl == 0) {
            bk.pend_key[rec_out0 + L.run[s]] = key;
            if constexpr (LOCATED) bk.pend_slot[rec_out0 + L.run[s]] = slot;
        }
    }
}

// Merge, one key with more records than a pass holds (a key that is a large share of a large batch: more than kSlab slabs each hold it): every
// tile adds the records j = its number, +32, +64, … of bucket records [beg, beg + R) that carry `key` into one fp64 partial row, the rows are
// combined after a barrier, the key is updated once.
template <int KIND, int DIM4, bool LOCATED, bool GROUPED = false>
__device__ __forceinline__ void mono_pass(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, uint32_t beg, uint32_t R, int64_t key, uint32_t any_rec,
                                          const GroupDesc* gdesc = nullptr) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15, wv = threadIdx.x >> 6;
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4, q = (uint32_t)wv * 4 + tile;
    constexpr uint32_t kTiles = kApplyWaves * 4;
    OptArgs a = A.a;
    a.kind = KIND;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t pb = atomicAdd(&A.op->n_part, kTiles);
        if (pb + kTiles > A.max_part) pb = 0u;   // (see process_slab)
        L.part_base = pb;
    }
    __syncthreads();
    for (uint32_t col = tl; col < dim4; col += 16) {
        double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
        for (uint32_t j0 = q; j0 < R; j0 += 2 * kTiles) {   // two records in flight; rows are read unconditionally (valid addresses), added only on a match
            double2 lo[2], hi[2];
            bool on[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t j = min(j0 + (uint32_t)u * kTiles, R - 1);
                on[u] = j0 + (uint32_t)u * kTiles < R && bk.pend_key[beg + j] == key;
                const double2* r = reinterpret_cast<const double2*>(bk.pend_row + ((uint64_t)(beg + j) * dim4 + col) * 4);
                lo[u] = r[0]; hi[u] = r[1];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { sx += on[u] ? lo[u].x : 0.0; sy += on[u] ? lo[u].y : 0.0; sz += on[u] ? hi[u].x : 0.0; sw += on[u] ? hi[u].y : 0.0; }
        }
        double2* dst = reinterpret_cast<double2*>(A.part + ((uint64_t)(L.part_base + q) * dim4 + col) * 4);
        dst[0] = make_double2(sx, sy); dst[1] = make_double2(sz, sw);
    }
    __syncthreads();   // the partial rows were written by this block: visible to its tile 0 after the barrier
    int64_t slot;
    if constexpr (LOCATED) slot = bk.pend_slot[beg + any_rec];   // every record of the key carries its slot
    else {
        bool is_new, full;
        slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, q == 0, tile, tl, is_new, full);
    }
    if (q == 0 && slot >= 0) {
        const RowAt at = row_at<GROUPED>(A, gdesc, slot, true);
        for (uint32_t col = tl; col < dim4; col += 16) {
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
#pragma unroll 2   // (fully unrolled, the 32 rows' loads held 128 VGPRs: this rare path must not set the kernel's register count)
            for (uint32_t u = 0; u < kTiles; ++u) {
                const double2* r = reinterpret_cast<const double2*>(A.part + ((uint64_t)(L.part_base + u) * dim4 + col) * 4);
                sx += r[0].x; sy += r[0].y; sz += r[1].x; sw += r[1].y;
            }
            const uint64_t o = at.row * dim4 + col;
            float4 w = at.values[o], x1 = at.s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
            opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
            store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
            if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
        }
    }
    __syncthreads();
}

// A bucket's entries lie in <= kPartBlocks runs, one per partition block, inside the partition blocks' slices of pos / pkey.  Lane l of wave 0
// fetches the lengths and places of runs 2l and 2l + 1 (seg_load: four independent loads), a wave scan turns the lengths into each run's first
// index within the bucket (seg_scan -> L.seg_first / L.seg_at; the caller's next barrier publishes them).
struct SegRuns { uint32_t len0, len1, at0, at1; };
__device__ __forceinline__ SegRuns seg_load(const ApplyArgs& A, const BucketScratch& bk, uint32_t b, uint32_t tid) {
    SegRuns r{0u, 0u, 0u, 0u};
    if (tid < 64) {
        const uint32_t k0 = 2 * tid, k1 = k0 + 1;
        if (k0 < A.part_blocks) { r.len0 = bk.cnt_mat[(uint64_t)k0 * A.nbk + b]; r.at0 = k0 * A.per_block + bk.off_mat[(uint64_t)k0 * A.nbk + b]; }
        if (k1 < A.part_blocks) { r.len1 = bk.cnt_mat[(uint64_t)k1 * A.nbk + b]; r.at1 = k1 * A.per_block + bk.off_mat[(uint64_t)k1 * A.nbk + b]; }
    }
    return r;
}
__device__ __forceinline__ void seg_scan(ApplyLds& L, const SegRuns& r, uint32_t tid) {
    if (tid >= 64) return;
    const uint32_t both = r.len0 + r.len1;
    const uint32_t incl = wave_incl_scan_u32(both);
    const uint32_t k0 = 2 * tid;
    L.seg_first[k0] = incl - both;
    L.seg_first[k0 + 1] = incl - r.len1;
    if (tid == 63) L.seg_first[kPartBlocks] = incl;
    L.seg_at[k0] = r.at0;
    L.seg_at[k0 + 1] = r.at1;
}

// The two roles of the apply kernel.  bucket_role: one block per bucket, for the buckets that fit one slab — all of them on a batch without
// hot keys.  split_role: the slabs of the buckets that do not, and their merges — few blocks with long chains (slab, hand-off, merge), so
// they lead the grid.  They are separate code paths of ONE kernel, each with its own instance of process_slab: the bucket role's needs 64
// VGPRs, the split role's 107 — one shared instance (emit / src_rec as run-time flags inside a loop) kept the rare path's state live in every
// block's hot path.  The kernel is bounded to 80 VGPRs (MEE_APPLY_WAVES = 6, meepo_apply_part.h: three 512-thread blocks per CU); what does
// not fit spills in the split role only (24-56 B per lane).  Tried instead: two kernels, the second launched with hipExtAnyOrderLaunch so
// that they share the device — the flag is not supported on gfx9, the launches serialise (Zipf(1.05): 48 + 39 us against 79 us); the split
// role as a __noinline__ function — a kernel's register allocation covers its callees (400-500 B of stack per lane).
template <int KIND, int DIM4, bool LOCATED, bool GROUPED>
__device__ __forceinline__ void bucket_role(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, const uint32_t b, const GroupDesc* gdesc) {
    // ONE round trip brings everything the block must know before it can fetch its entries: which copy of the totals this batch's partition
    // filled (bk.seq[1], meepo_apply_part.h), the bucket's total in BOTH copies, and — lanes of wave 0 — the lengths and places of the bucket's
    // runs in the partition blocks' slices.  (As a chain seq -> total -> run lengths these were three dependent loads, 2-3 us of every block's
    // life before its first useful request.)
    MEE_TL(A, 0);
    const SegRuns runs = seg_load(A, bk, b, threadIdx.x);
    const uint32_t tot0 = bk.tot[b], tot1 = bk.tot[bk.n_buckets_max + b];
    const uint32_t parity = bk.seq[1];
    if (b == 0 && threadIdx.x == 0) atomicAdd(&bk.seq[0], 1u);   // this partition is consumed: the next one fills the other copy
    const uint32_t size = __builtin_amdgcn_readfirstlane(parity ? tot1 : tot0);
    if (size == 0 || size > kSlab) return;   // an empty bucket | a split bucket: the spare blocks (split_role) have it
    MEE_TL(A, 1);   // first round trip done (size known)
#if MEE_APPLY_TIMELINE
    if (threadIdx.x == 0) { A.dbg[(uint64_t)blockIdx.x * 8 + 6] = size | (unsigned long long)b << 32; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); A.dbg[(uint64_t)blockIdx.x * 8 + 7] = (unsigned long long)(xcc & 0xf) << 32 | hw; }
#endif
    seg_scan(L, runs, threadIdx.x);
    process_slab<KIND, DIM4, LOCATED, false, GROUPED>(L, A, bk, 0, size, false, false, b, 0, gdesc);
    MEE_TL(A, 5);
}

template <int KIND, int DIM4, bool LOCATED, bool GROUPED>
__device__ __forceinline__ void split_role(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, const uint32_t e0, const uint32_t stride, const GroupDesc* gdesc) {
    // Block e takes slabs e, e + stride, … of the buckets that hold more than one slab (a hot key), found by a prefix sum over the bucket
    // totals — and leaves at once when the partition saw no such bucket.
    const uint32_t hs0 = bk.has_split[0], hs1 = bk.has_split[1];
    const uint32_t parity = __builtin_amdgcn_readfirstlane(bk.seq[1]);   // which copy of the totals this batch's partition filled
    if (!__builtin_amdgcn_readfirstlane(parity ? hs1 : hs0)) return;
    const uint32_t tot_base = parity * bk.n_buckets_max;
    for (uint32_t e = e0;; e += stride) {
        // (the thread index is re-read through an empty asm in every turn, as in process_slab: hoisted out of this loop, the per-thread addresses
        // of everything below stayed live across the whole role and were spilled to scratch — a memory round trip in front of every use)
        uint32_t tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        uint32_t b, sub;
        {
            const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads;
            unsigned long long mine = 0;
            for (uint32_t q = 0; q < per_t; ++q) {
                const uint32_t bb = tx * per_t + q;
                const uint32_t tt = bb < A.nbk ? bk.tot[tot_base + bb] : 0u;
                mine += tt > kSlab ? (tt + kSlab - 1) / kSlab : 0u;
            }
            unsigned long long total;
            const uint32_t ex = (uint32_t)block_scan_u64<kApplyWaves>(mine, L.wsum, total);
            if (e >= (uint32_t)total) return;   // block-uniform
            if (e >= ex && e < ex + (uint32_t)mine) {   // exactly one thread
                uint32_t acc = ex;
                for (uint32_t q = 0; q < per_t; ++q) {
                    const uint32_t bb = tx * per_t + q;
                    const uint32_t tt = bb < A.nbk ? bk.tot[tot_base + bb] : 0u;
                    const uint32_t x = tt > kSlab ? (tt + kSlab - 1) / kSlab : 0u;
                    if (e < acc + x) { L.rec_base = bb; L.n_cand = e - acc; break; }
                    acc += x;
                }
            }
            __syncthreads();
            b = __builtin_amdgcn_readfirstlane(L.rec_base); sub = __builtin_amdgcn_readfirstlane(L.n_cand);
            __syncthreads();
        }
        const uint32_t size = __builtin_amdgcn_readfirstlane(bk.tot[tot_base + b]);
        seg_scan(L, seg_load(A, bk, b, tx), tx);   // the bucket's runs in the partition blocks' slices
        uint32_t beg;   // the bucket's first pending record = the keys in the buckets before it (records never outnumber positions)
        {
            unsigned long long mine = 0, total;
            for (uint32_t bb = tx; bb < b; bb += kApplyThreads) mine += bk.tot[tot_base + bb];
            (void)block_scan_u64<kApplyWaves>(mine, L.wsum, total);
            beg = __builtin_amdgcn_readfirstlane((uint32_t)total);
            __syncthreads();
        }
        // the slab first; in the bucket's LAST slab the same loop then runs the merge passes (src_rec)
        bool merging = false;
        uint32_t first = sub * kSlab, m = min(kSlab, size - sub * kSlab), R = 0, bits0 = 0;
        uint64_t v0 = 0;
        bool done = false;
        while (!done) {
            process_slab<KIND, DIM4, LOCATED, true, GROUPED>(L, A, bk, first, m, !merging, merging, b, beg, gdesc);
            if (!merging) {
                // ---- publish this slab's pending records, take a ticket; the slab that draws the last ticket merges the bucket.  (The in-launch
                // hand-off of cdna_hip_programming.md Guideline 16 in its counter form: plain stores, every wave drains them, one lane releases at
                // agent scope, THEN the ticket; the last arriver acquires at agent scope before any of its waves reads a record.  No block ever
                // waits for another.)
                const uint32_t nsub = (size + kSlab - 1) / kSlab;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const uint32_t tk = __hip_atomic_fetch_add(&bk.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    L.is_last = tk == nsub - 1;
                    if (tk == nsub - 1) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __syncthreads();
                if (!L.is_last) break;   // block-uniform: on to this block's next slab
                // Merge: records of one key must meet in one pass and a pass holds kSlab records, so the records are taken by the low bits of
                // mix64b(key): `bits0` bits give passes of ~256 records; a pass that still finds more than kSlab splits on one more bit, and a
                // pass whose records all carry ONE key goes to mono_pass (mix64b is a bijection: the splitting ends).
                merging = true;
                R = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&bk.pend_cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // every slab added its runs before its ticket
                while (((uint64_t)kSlab / 2 << bits0) < R && bits0 < 40) ++bits0;
                v0 = 0;
                if (threadIdx.x == 0) L.stk_n = 0u;
                first = 0;
            }
            // ---- the next merge pass: pop a hash prefix, collect its records; too many -> split the prefix (or one key: mono_pass) ----
            m = 0;
            while (m == 0 && !done) {   // block-uniform
                __syncthreads();
                if (L.stk_n == 0) {
                    if (v0 >> bits0) { done = true; break; }   // every prefix done
                    __syncthreads();
                    if (threadIdx.x == 0) { L.stk_n = 1u; L.stk_bits[0] = bits0; L.stk_val[0] = v0; }
                    ++v0;
                    __syncthreads();
                }
                const uint32_t top = __builtin_amdgcn_readfirstlane(L.stk_n - 1), bits = __builtin_amdgcn_readfirstlane(L.stk_bits[top]);
                const unsigned long long val_v = L.stk_val[top];
                const uint64_t val = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)val_v) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(val_v >> 32)) << 32;
                const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
                __syncthreads();
                if (threadIdx.x == 0) { L.stk_n = top; L.n_cand = 0u; L.kmin = ~0ull; L.kmax = 0ull; }
                __syncthreads();
                for (uint32_t j = threadIdx.x; j < R; j += kApplyThreads) {
                    const int64_t kj = bk.pend_key[beg + j];
                    if ((mix64b((uint64_t)kj) & mask) != val) continue;
                    const uint32_t q = atomicAdd(&L.n_cand, 1u);
                    if (q < kSlab) L.src[q] = j;
                    atomicMin(&L.kmin, (unsigned long long)kj ^ kBias);
                    atomicMax(&L.kmax, (unsigned long long)kj ^ kBias);
                }
                __syncthreads();
                const uint32_t nc = __builtin_amdgcn_readfirstlane(L.n_cand);
                if (nc > kSlab) {
                    if (L.kmin == L.kmax) {   // more records of ONE key than a pass holds
                        mono_pass<KIND, DIM4, LOCATED, GROUPED>(L, A, bk, beg, R, (int64_t)(L.kmin ^ kBias), L.src[0], gdesc);
                    } else if (threadIdx.x == 0 && bits < 64 && L.stk_n + 2 <= 72) {
                        L.stk_bits[L.stk_n] = bits + 1; L.stk_val[L.stk_n] = val; ++L.stk_n;
                        L.stk_bits[L.stk_n] = bits + 1; L.stk_val[L.stk_n] = val | 1ull << bits; ++L.stk_n;
                    }
                } else m = nc;
            }
        }
    }   // next slab of this block
}

constexpr uint32_t kGroupDescLds = 64;   // members whose descriptors a GROUPED kernel stages in LDS (3 KB): larger groups read them from device memory
template <int KIND, int DIM4, bool LOCATED, bool GROUPED = false>
__global__ __launch_bounds__(kApplyThreads, MEE_APPLY_WAVES) void bkt_apply_kernel(ApplyArgs A, BucketScratch bk) {
    __shared__ ApplyLds L;
    const GroupDesc* gdesc = nullptr;
    if constexpr (GROUPED) {   // the members' planes: needed once per work item, so they come out of LDS (the first barrier inside the roles publishes them)
        __shared__ GroupDesc gd[kGroupDescLds];
        if (A.n_tables <= kGroupDescLds) {
            for (uint32_t j = threadIdx.x; j < A.n_tables; j += kApplyThreads) gd[j] = A.desc[j];
            gdesc = gd;
        }
    }
    if (blockIdx.x < A.n_extra) split_role<KIND, DIM4, LOCATED, GROUPED>(L, A, bk, blockIdx.x, A.n_extra, gdesc);   // block-uniform
#if MEE_APPLY_TIMELINE == 2   // diagnostic: decouple a bucket's parity from its block's XCD (blocks are dealt round-robin over the 8 XCDs)
    else { const uint32_t i = blockIdx.x - A.n_extra; bucket_role<KIND, DIM4, LOCATED, GROUPED>(L, A, bk, (i ^ ((i >> 3) & 1u)) < A.nbk ? i ^ ((i >> 3) & 1u) : i, gdesc); }
#else
    else bucket_role<KIND, DIM4, LOCATED, GROUPED>(L, A, bk, blockIdx.x - A.n_extra, gdesc);
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
int bucket_scratch_alloc(mee_table* t) {
    BucketScratch& bk = t->bk;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device);
    bk.slots = (uint32_t)cus * kApplyBlocksPerCU;
    const uint64_t most = (uint64_t)(kMaxBuckets / bk.slots ? kMaxBuckets / bk.slots * bk.slots : kMaxBuckets) * kBucketMax;   // beyond that buckets would outgrow their slabs
    bk.fast_max = t->max_batch < most ? t->max_batch : most;
    bk.n_buckets_max = kMaxBuckets;
    bk.pos = t->bs.occ;   // max_batch entries; the group-table apply and this one never run at the same time on one table
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, uint64_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) t->workspace_bytes += bytes; } };
    alloc((void**)&bk.pkey, bk.fast_max * 8);
    alloc((void**)&bk.cnt_mat, (uint64_t)kPartBlocks * bk.n_buckets_max * 4);
    alloc((void**)&bk.off_mat, (uint64_t)kPartBlocks * bk.n_buckets_max * 4);
    alloc((void**)&bk.tot, 2ull * bk.n_buckets_max * 4);
    alloc((void**)&bk.has_split, 2 * 4);
    alloc((void**)&bk.seq, 2 * 4);
    if (e == hipSuccess) e = hipMemset(bk.seq, 0, 2 * 4);
    if (e == hipSuccess) e = hipMemset(bk.tot, 0, 2ull * bk.n_buckets_max * 4);   // (every partition launch zeroes the copy the next one adds to)
    if (e == hipSuccess) e = hipMemset(bk.has_split, 0, 2 * 4);
    alloc((void**)&bk.pend_cnt, (uint64_t)bk.n_buckets_max * 4);
    alloc((void**)&bk.ticket, (uint64_t)bk.n_buckets_max * 4);
    alloc((void**)&bk.pend_key, bk.fast_max * 8);
    alloc((void**)&bk.pend_slot, bk.fast_max * 8);
    alloc((void**)&bk.pend_row, bk.fast_max * (uint64_t)t->dim * sizeof(double));
    if (e != hipSuccess) return fail(MEE_ERR_OUT_OF_MEMORY, "hipMalloc for the apply scratch: %s", hipGetErrorString(e));
    return MEE_OK;
}
void bucket_scratch_free(mee_table* t) {
    BucketScratch& bk = t->bk;
    void* dev[] = {bk.pkey, bk.cnt_mat, bk.off_mat, bk.tot, bk.has_split, bk.seq, bk.pend_cnt, bk.ticket, bk.pend_key, bk.pend_slot, bk.pend_row};
    for (void* p : dev) if (p) (void)hipFree(p);
}

#if MEE_APPLY_TIMELINE
static unsigned long long* g_dbg = nullptr;
extern "C" int mee_debug_timeline(unsigned long long* host_out, uint64_t n_words) {
    if (!g_dbg) return 1;
    return hipMemcpy(host_out, g_dbg, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif
int bucket_apply_prepare(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st) {
    const uint32_t nbk = bucket_count_for(t, n);
    uint32_t blocks, per_block;
    part_geometry(n, kPartThreads, blocks, per_block);
    t->part_blocks = blocks; t->part_per_block = per_block;
    bkt_sort_kernel<<<blocks, kPartThreads, nbk * 4, st>>>(d_keys, n, nbk, per_block, t->bk, &t->ctr->status, t->op);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// a prepared partition that no apply will consume (mee_apply_discard): count it as consumed, so that the next partition fills the other copy
__global__ void bkt_skip_kernel(BucketScratch bk) { if (threadIdx.x == 0) bk.seq[0] += 1u; }
int bucket_apply_discard(mee_table* t, hipStream_t st) {
    bkt_skip_kernel<<<1, 64, 0, st>>>(t->bk);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

int bucket_apply_launch(mee_table* t, const float* d_grads, uint32_t n, const OptArgs& a, const uint32_t* d_gidx, const int64_t* d_slots, hipStream_t st,
                        const GroupDesc* d_desc, uint32_t n_tables) {
    ApplyArgs A{};
    A.tkeys = t->keys; A.values = (float4*)t->values; A.s1 = (float4*)t->s1; A.s2 = (float4*)t->s2; A.nb = t->nb; A.dim4 = t->dim4;
    A.grads = (const float4*)d_grads; A.gidx = d_gidx; A.slots = d_slots;
    A.capacity = t->capacity; A.handle_tag = (int64_t)(t->handle_epoch & kHandleEpochMask) << kHandleSlotBits; A.status = &t->ctr->status;
    A.part = t->bs.gacc; A.max_part = t->bs.max_part; A.op = t->op; A.a = a;
#if MEE_APPLY_TIMELINE
    if (!g_dbg) { (void)hipMalloc((void**)&g_dbg, 16384 * 8 * 8); (void)hipMemset(g_dbg, 0, 16384 * 8 * 8); }
    A.dbg = g_dbg;
#endif
    A.desc = d_desc; A.n_tables = n_tables;   // a table group's apply (d_slots = the batch's located rows = its keys; t = the group's scratch table)
    A.nbk = bucket_count_for(t, n);
    A.part_blocks = t->part_blocks; A.per_block = t->part_per_block;   // as whoever partitioned this batch left them
    // blocks [0, n_extra): the slabs of split buckets (few blocks, long chains: slab, hand-off, merge — so they lead the grid), one per CU unless
    // the batch cannot have that many slabs; block e takes slabs e, e + n_extra, …  (One block per POSSIBLE slab — 2n / 512 — put a thousand blocks
    // that only look at one word in front of every uniform batch: an apply alone 113 us instead of 97 us.)
    const uint32_t spare = t->bk.spare_blocks ? t->bk.spare_blocks : t->bk.slots / kApplyBlocksPerCU;
    A.n_extra = max_extra_slabs(n) < spare ? max_extra_slabs(n) : spare;
#define BKT(K, D4, LOC) bkt_apply_kernel<K, D4, LOC><<<A.n_extra + A.nbk, kApplyThreads, 0, st>>>(A, t->bk)
#define BKT_L(K, D4) do { if (d_desc) bkt_apply_kernel<K, D4, true, true><<<A.n_extra + A.nbk, kApplyThreads, 0, st>>>(A, t->bk); else if (d_slots) BKT(K, D4, true); else BKT(K, D4, false); } while (0)
#define BKT_D(K) do { if (t->dim4 == 16) BKT_L(K, 16); else if (t->dim4 == 32) BKT_L(K, 32); else BKT_L(K, 0); } while (0)
    if (a.kind == MEE_OPT_ADAGRAD) BKT_D(MEE_OPT_ADAGRAD); else BKT_D(MEE_OPT_ADAM);
#undef BKT_D
#undef BKT_L
#undef BKT
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

}  // namespace mee

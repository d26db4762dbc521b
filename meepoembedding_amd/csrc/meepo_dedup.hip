// meepo_dedup.hip — duplicate-key elimination and last-wins elections on the bucketed machinery (SPEC.md §3 "duplicates: last wins", §4
// "duplicate keys"): the partition of meepo_apply_part.h, then ONE kernel whose blocks find a bucket's distinct keys in an LDS hash table.
//
// Reference anchor: /root/reference/README.md:2 (no code upstream); BASELINE.json north_star "LDS-staged key buckets and wavefront ballot /
// prefix-sum for duplicate-key reduction".
//
// Rounds 2-3 did both through a GLOBAL group table: one scattered CAS per key to claim an entry (13-20 G scattered atomics/s chip-wide:
// 23 us per 256K keys, 90 us per 1M), a plan pass, a fill pass and an emit pass — 0.23 ms per 1M keys for mee_dedup_keys, which sits on the
// critical path of a sharded lookup with pre-exchange dedup, and 85 of assign's 227 us.  Here all occurrences of a key are in one bucket
// (the partition sorts the batch by the top bits of mix64(key), LDS histograms, no per-key global atomic), a 256-thread block takes a bucket
// of ANY size — there are no rows to sum, so no slabs and no pending records: a bucket larger than the LDS table is taken in passes by the
// low bits of mix64b(key) —, and the only global atomic is one per (block, pass) that reserves the pass's slice of the unique list.
//
//   bkt_dedup_keys_kernel   mee_dedup_keys: the distinct keys into d_uniq (each bucket's into its own slice of the list: EMPTY between them),
//                           each position's index into d_inverse; sync-free, no global atomic (the partition's launch pre-fills d_uniq with
//                           EMPTY and gives reserved keys their miss_index).
//   bkt_assign_kernel       mee_assign: per distinct key the LAST position wins (atomicMax in the LDS table), one probe per distinct key, the
//                           winner's row overwrites the key's row; every occurrence gets the key's found byte.
//   bkt_dedup_sum_kernel    mee_dedup_sum (round 5): dedup_keys' outputs plus, per distinct key, its occurrence count and the fp64 sum of its rows
//                           rounded once to fp32 — what a rank sends to a key's owner instead of one gradient row per occurrence.  Sync-free.
#include <hip/hip_runtime.h>

#include "meepo_apply_part.h"

namespace mee {

// The partition's geometry is the apply's (bucket_count_for: 3072 buckets of 341 per 1M keys, 128 partition blocks).  Measured alternatives, DESIGN.md §8: a bucket count
// by this file's own block slots (2048 buckets of <= 512: one round) — dedup_keys 70.9 -> 75.7 us per 1M keys —, 256 partition blocks — 70.2 -> 75.6 us (twice the runs per bucket).
constexpr int kDedupThreads = 256;
constexpr int kDedupWaves = kDedupThreads / 64;
constexpr uint32_t kDedupSlots = 1024;        // LDS hash table of one pass
constexpr uint32_t kDedupFill = 896;          // distinct keys a pass may hold (load 0.875); beyond: the pass is split by one more hash bit

// MEE_SUM_TIMELINE (diagnostic builds only: tools/sum_timeline.py): thread 0 of every block stamps the 100 MHz wall clock at its phase boundaries
#ifndef MEE_SUM_TIMELINE
#define MEE_SUM_TIMELINE 0
#endif
#if MEE_SUM_TIMELINE
__device__ unsigned long long* g_sum_dbg = nullptr;
#define MEE_STL(i) do { if (threadIdx.x == 0 && g_sum_dbg && blockIdx.x < 8192) g_sum_dbg[(uint64_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define MEE_STL(i) do { } while (0)
#endif

struct DedupLds {
    unsigned long long key[kDedupSlots];      // key ^ kBias, 0 = empty
    uint32_t val[kDedupSlots];                // dedup: the key's index in this pass's slice of the unique list | assign: 1 + the key's last position, later its found flag
    uint32_t cnt[kDedupSlots];                // occurrences (hot-key report)
    uint32_t seg_first[kPartBlocksMax + 1], seg_at[kPartBlocksMax];
    unsigned long long wsum[kDedupWaves];
    unsigned long long stk_val[72];           // passes still to do: (hash bits, value) — prefixes of mix64b(key), up to all 64 bits
    uint32_t stk_bits[72];
    uint32_t hp[8];                           // hot_plan's hand-over words; the hot windows of an assign
    uint32_t pre[kDedupWaves];                // per-wave partial results (the totals in front of a bucket; a window's last position)
    uint32_t stk_n, n_distinct, overflow, base;
};

__device__ __forceinline__ uint32_t dd_entry_at(const DedupLds& L, uint32_t gi) {   // index within the bucket -> its place in pos / pkey
    uint32_t k = 0;
#pragma unroll
    for (uint32_t stp = kPartBlocksMax / 2; stp; stp >>= 1) if (L.seg_first[k + stp] <= gi) k += stp;
    return L.seg_at[k] + (gi - L.seg_first[k]);
}

// the bucket's runs in the partition blocks' slices (wave 0: two runs per lane), as in the apply kernel: the loads (dd_seg_load) travel with the
// kernel's first round trip, the scan (dd_seg_scan) puts the runs' first indices into LDS
constexpr uint32_t kRunsPerLane = kPartBlocksMax / 64;
struct DdRuns { uint32_t len[kRunsPerLane], at[kRunsPerLane]; };   // lane l of wave 0: runs kRunsPerLane * l ..
__device__ __forceinline__ DdRuns dd_seg_load(const BucketScratch& bk, uint32_t nbk, uint32_t part_blocks, uint32_t per_block, uint32_t b, uint32_t tid) {
    DdRuns r{};
    if (tid < 64) {
#pragma unroll
        for (uint32_t q = 0; q < kRunsPerLane; ++q) {
            const uint32_t k = kRunsPerLane * tid + q;
            if (k < part_blocks) { r.len[q] = bk.cnt_mat[(uint64_t)k * nbk + b]; r.at[q] = k * per_block + bk.off_mat[(uint64_t)k * nbk + b]; }
        }
    }
    return r;
}
__device__ __forceinline__ void dd_seg_scan(DedupLds& L, const DdRuns& r, uint32_t tid) {
    if (tid >= 64) return;
    uint32_t all = 0;
#pragma unroll
    for (uint32_t q = 0; q < kRunsPerLane; ++q) all += r.len[q];
    const uint32_t incl = wave_incl_scan_u32(all);
    uint32_t first = incl - all;
#pragma unroll
    for (uint32_t q = 0; q < kRunsPerLane; ++q) { L.seg_first[kRunsPerLane * tid + q] = first; L.seg_at[kRunsPerLane * tid + q] = r.at[q]; first += r.len[q]; }
    if (tid == 63) L.seg_first[kPartBlocksMax] = incl;
}

// The table is probed by DOUBLE hashing — first slot and (odd) stride from the TOP bits of mix64b(key); the passes split on its low bits —: a hash bucket of ~683
// positions fills the 1024 slots to 0.67, where linear probing's clusters made the slowest of a wave's 64 lanes probe 30-40 times (the wave waits for it: 12 us of a
// block's build, 5 of its look-ups); without clusters the slowest lane of a wave needs ~12.
__device__ __forceinline__ uint32_t dd_slot0(uint64_t h) { return (uint32_t)(h >> 54); }
__device__ __forceinline__ uint32_t dd_stride(uint64_t h) { return ((uint32_t)(h >> 43) & (kDedupSlots - 1)) | 1u; }
static_assert(kDedupSlots == 1024, "dd_slot0 takes the top 10 bits");
// slot of `bkey` in the pass's table (the key is there)
__device__ __forceinline__ uint32_t dd_lookup(const DedupLds& L, unsigned long long bkey) {
    const uint64_t h = mix64b(bkey ^ kBias);
    uint32_t s = dd_slot0(h);
    const uint32_t stride = dd_stride(h);
    while (L.key[s] != bkey) s = (s + stride) & (kDedupSlots - 1);
    return s;
}

// One pass over the bucket: every entry whose key has `val` in the low `bits` bits of mix64b goes into the LDS table.  LAST: the table keeps
// 1 + the key's highest batch position.  A wave whose 64 entries all carry ONE key (a hot key's own bucket, 15 000 entries of one key) inserts
// it once.  Returns false if the pass holds more distinct keys than the table takes (the caller splits it on one more bit).
// A bucket of up to 4 x 256 entries — nearly every hash bucket — is held in registers from the build pass on (batch position of the thread's four entries and the
// table slot each one's key went to; kNoSlot: not an entry of this pass): the pass that writes each position's result afterwards neither fetches them a second time
// (one dependent round trip less per block) nor looks their keys up again (a hash and a probe sequence per entry).
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
struct DdHeld { uint32_t s[4]; uint32_t p[4]; bool valid; };
template <bool LAST>
__device__ __forceinline__ bool dd_build(DedupLds& L, const BucketScratch& bk, uint32_t size, uint32_t bits, uint64_t val, DdHeld& H) {
    const uint32_t t = threadIdx.x;
    H.valid = size <= 4u * kDedupThreads;   // (block-uniform)
    for (uint32_t j = t; j < kDedupSlots; j += kDedupThreads) { L.key[j] = 0ull; L.val[j] = 0u; L.cnt[j] = 0u; }
    if (t == 0) { L.n_distinct = 0u; L.overflow = 0u; }
    const bool may_overflow = size > kDedupFill;   // (block-uniform) a pass over at most kDedupFill entries cannot: nobody counts its keys
    lds_barrier();
    MEE_STL(7);   // table cleared
    const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
    constexpr int kIn = 4;   // entries a thread has in flight per step (one dependent round trip per step: a bucket of 4 000 entries takes 4 steps, not 16)
    for (uint32_t e0 = 0; e0 < size; e0 += kIn * kDedupThreads) {   // block-uniform trip count: the wave ballots below need whole waves
        int64_t kq[kIn];
        uint32_t pq[kIn];
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const uint32_t e = e0 + (uint32_t)q * kDedupThreads + t;
            kq[q] = kEmpty; pq[q] = 0u;
            if (e < size) {
                const uint32_t at = dd_entry_at(L, e);
                const PartEntry en = bk.ent[at];
                kq[q] = en.key;
                if (LAST || H.valid) pq[q] = en.pos;
            }
        }
        if (H.valid) {
#pragma unroll
            for (int q = 0; q < kIn; ++q) { H.s[q] = kNoSlot; H.p[q] = pq[q]; }
        }
#if MEE_SUM_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MEE_STL(8);   // thread 0's entries are in
#endif
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int64_t key = kq[q];
            const uint32_t p = pq[q];
            const unsigned long long bkey = (unsigned long long)key ^ kBias;
            const bool mine = key != kEmpty && (mix64b((uint64_t)key) & mask) == val;   // (EMPTY is in no bucket: it marks a lane past the end)
            // all of the wave's entries in this pass carry one key: one lane speaks for the wave
            const unsigned long long act = __ballot(mine);
            if (act == 0ull) continue;
            const int lead = __ffsll((long long)act) - 1;
            const unsigned long long k0 = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(bkey >> 32), lead) << 32) | (uint32_t)__shfl((int)(uint32_t)bkey, lead);
            const bool uniform = __ballot(mine && bkey == k0) == act;
            uint32_t pmax = mine ? p : 0u;
            if (LAST && uniform) {   // the wave's highest position of that key
#pragma unroll
                for (int d = 32; d; d >>= 1) pmax = max(pmax, (uint32_t)__shfl_xor((int)pmax, d));
            }
            uint32_t my_s = kNoSlot;
            if (mine && (!uniform || (int)(t & 63) == lead) && (!may_overflow || __hip_atomic_load(&L.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)) {
                const uint64_t h = mix64b((uint64_t)key);
                uint32_t s = dd_slot0(h);
                const uint32_t stride = dd_stride(h);
                bool placed = false;
                for (uint32_t probes = 0; probes < kDedupSlots; ++probes) {   // (bounded: a table that other threads fill up meanwhile must not trap this one; an odd stride visits every slot)
                    // a plain read first: an occupied slot (another key: next slot; this key: done) costs no compare-and-swap
                    unsigned long long old = __hip_atomic_load(&L.key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old == 0ull) old = atomicCAS(&L.key[s], 0ull, bkey);
                    if (old == 0ull) {
                        if (may_overflow && atomicAdd(&L.n_distinct, 1u) >= kDedupFill) __hip_atomic_store(&L.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        placed = true; break;
                    }
                    if (old == bkey) { placed = true; break; }
                    s = (s + stride) & (kDedupSlots - 1);
                }
                if (placed) {
                    atomicAdd(&L.cnt[s], uniform ? (uint32_t)__popcll(act) : 1u);
                    if (LAST) atomicMax(&L.val[s], 1u + pmax);
                    my_s = s;
                } else __hip_atomic_store(&L.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (H.valid) {   // (block-uniform; every lane of the wave is here: `act` is the same in all of them)
                if (uniform) my_s = (uint32_t)__shfl((int)my_s, lead);   // one lane spoke for the wave
                H.s[q] = mine ? my_s : kNoSlot;
            }
        }
    }
    lds_barrier();
    return L.overflow == 0u;
}

// the pass driver: DFS over hash prefixes, every pass that fits is handed to `emit(bits, val)`
// (The prefixes are 64 bits wide: mix64b is a bijection, so ANY set of distinct keys separates within 64 bits and the stack never holds more than 65 entries.
// Round 4 stopped at 32 bits and dropped — silently — a pass of more than 896 keys that share the low 32 bits of mix64b: constructible, both mixers are public.)
template <bool LAST, class Emit>
__device__ __forceinline__ void dd_passes(DedupLds& L, const BucketScratch& bk, uint32_t size, uint32_t parity, uint32_t hot_count, uint32_t* status, Emit emit) {
    if (threadIdx.x == 0) { L.stk_n = 1u; L.stk_bits[0] = 0u; L.stk_val[0] = 0ull; }
    lds_barrier();
    while (true) {
        const uint32_t n = __builtin_amdgcn_readfirstlane(L.stk_n);
        if (n == 0) break;
        const uint32_t bits = __builtin_amdgcn_readfirstlane(L.stk_bits[n - 1]);
        const unsigned long long val_v = L.stk_val[n - 1];
        const uint64_t val = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)val_v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(val_v >> 32)) << 32;
        lds_barrier();
        if (threadIdx.x == 0) L.stk_n = n - 1;
        DdHeld held;
        if (dd_build<LAST>(L, bk, size, bits, val, held)) {
            // keys with enough occurrences to fill half a slab get a bucket of their own in the next batch (meepo_apply_part.h)
            for (uint32_t s = threadIdx.x; s < kDedupSlots; s += kDedupThreads)
                if (L.cnt[s] >= hot_count) report_hot_key(bk, parity, (int64_t)(L.key[s] ^ kBias), hot_cap_of(L.cnt[s], hot_count));
            emit(bits, val, held);
        } else if (threadIdx.x == 0) {
            if (bits < 64 && L.stk_n + 2 <= 72) {
                const uint32_t m = L.stk_n;
                L.stk_bits[m] = bits + 1; L.stk_val[m] = val;
                L.stk_bits[m + 1] = bits + 1; L.stk_val[m + 1] = val | 1ull << bits;
                L.stk_n = m + 2;
            } else if (status) atomicOr(status, (uint32_t)MEE_STATUS_INTERNAL);   // cannot happen (see above); never lose keys silently
        }
        lds_barrier();
    }
}

struct DedupArgs {
    uint32_t nbk, nbk_hash, part_blocks, per_block, hot_count;
    OpCounters* op;
    uint32_t* h_slabs;
    int64_t* uniq; int64_t* inverse;   // dedup
    uint32_t size_from_runs;           // the bucket totals were not computed (assign without hot keys' buckets): a bucket's size is the sum of its runs
    uint32_t* status;                  // the table's sticky status word
    uint32_t hash_first;               // grid order (dd_unit): the hot keys' buckets (one block each: the key's report), the hash buckets, the hot keys' windows LAST
};

// A hot key's own bucket (b >= nbk_hash: the partition sent exactly ONE key there) needs no table: every entry is that key.  It is cut into
// WINDOWS of kHotWindow entries, one block each — the blocks behind the buckets in the grid —, so that a key with 80 000 occurrences in a
// batch of 1M is 20 blocks' work, not one block's 300 dependent steps.  What the windows of a bucket must agree on costs no communication: the
// hot keys' numbers in the unique list come FIRST — hot bucket h gets the count of non-empty hot buckets in front of it, which every block
// reads off the same kHotCap totals —, and the hash buckets' blocks number their keys from H (the non-empty hot buckets) on.
constexpr uint32_t kHotWindow = 4096;
constexpr uint32_t kSumWindow = 1024;   // ... of a dedup that also SUMS rows (every entry of a window is a 256-byte row to fetch: 16 round trips for the block's 16 tiles)
struct HotPlan { uint32_t H, units, h, win, n_win, rank, size; bool valid; };   // H, units: non-empty hot buckets and their windows in all;   // this block's window: hot bucket h, window `win` of n_win, the key's number `rank`
template <uint32_t WIN>
__device__ __forceinline__ HotPlan hot_plan(DedupLds& L, const BucketScratch& bk, const DedupArgs& A, uint32_t parity_guess_tot0, uint32_t parity_guess_tot1, uint32_t parity, int x /* window unit, or -1: only H is wanted */) {
    // (called by all threads; thread h < n_hot <= 128 holds hot bucket h's total.)  Waves 0 and 1 scan their 64 buckets' window counts and non-empty
    // flags (DPP), LDS carries wave 0's sums to wave 1, the thread whose bucket holds window x publishes it: a dozen instructions per thread
    // instead of every thread walking all 128 totals (2-3 us of every block's life on a skewed stream).
    const uint32_t n_hot = A.nbk - A.nbk_hash;
    const uint32_t t = threadIdx.x;
    HotPlan P{0, 0, 0, 0, 0, 0, 0, false};
    if (n_hot == 0) return P;   // (block-uniform)
    const uint32_t sz = t < n_hot ? (parity ? parity_guess_tot1 : parity_guess_tot0) : 0u;
    const uint32_t nw = (sz + WIN - 1) / WIN, ne = sz != 0;
    uint32_t iw = 0, ie = 0;
    if (t < 128) { iw = wave_incl_scan_u32(nw); ie = wave_incl_scan_u32(ne); }
    if (t == 63) { L.hp[0] = iw; L.hp[1] = ie; }
    if (t == 0) L.hp[6] = 0xFFFFFFFFu;
    lds_barrier();
    if (t >= 64 && t < 128) { iw += L.hp[0]; ie += L.hp[1]; }
    if (t == 127) { L.hp[2] = iw; L.hp[3] = ie; }
    if (x >= 0 && t < 128 && nw != 0 && (uint32_t)x >= iw - nw && (uint32_t)x < iw) {   // exactly one thread (the windows' ranges are disjoint)
        L.hp[4] = (uint32_t)x - (iw - nw); L.hp[5] = nw; L.hp[6] = t; L.hp[7] = ie - 1u; L.base = sz;
    }
    lds_barrier();
    P.units = L.hp[2]; P.H = L.hp[3];
    if (x >= 0 && L.hp[6] != 0xFFFFFFFFu) { P.win = L.hp[4]; P.n_win = L.hp[5]; P.h = L.hp[6]; P.rank = L.hp[7]; P.size = L.base; P.valid = true; }
    lds_barrier();   // (the hand-over words are reused)
    return P;
}

// which unit a block takes (unit numbers: buckets [0, nbk) — hash buckets, then the hot keys' own —, windows from nbk on).  dedup_keys / assign: the windows of the
// hot keys' buckets — their longest units: 4096 entries each — are the FIRST blocks of the grid, the buckets follow.  hash_first (dedup_sum: a hash bucket is ~80 us
// of a block's life, a window of 1024 rows ~35): the hash buckets start with the kernel and the windows fill the slots they free — longest units first; in front of them
// the blocks of the hot keys' buckets, whose only job is to report their key for the next batch BEFORE the hash buckets' blocks report theirs (the set admits first
// comers: a listed key must not lose its number to a newcomer of its own size).
__device__ __forceinline__ uint32_t dd_unit(const DedupArgs& A) {
    if (A.hash_first) {
        const uint32_t n_hot = A.nbk - A.nbk_hash;
        return blockIdx.x < n_hot ? A.nbk_hash + blockIdx.x : blockIdx.x < A.nbk ? blockIdx.x - n_hot : blockIdx.x;
    }
    const uint32_t w = gridDim.x - A.nbk;
    return blockIdx.x < w ? A.nbk + blockIdx.x : blockIdx.x - w;
}

// the block's bucket: (size, parity) — ONE round trip brings the bucket's totals (both copies), the copy selector, the hot keys' totals, the
// bucket's runs in the partition blocks' slices and (PREFIX: the dedup) the totals of the hash buckets in front of this one.  `P.H` = the
// non-empty hot keys' buckets (their keys are numbered first); a block beyond the buckets (blockIdx >= nbk) gets its window of a hot key's
// bucket in `P` instead.  `before` (PREFIX) = the positions in the hash buckets in front of this one.
template <bool PREFIX, uint32_t WIN = kHotWindow>
__device__ __forceinline__ uint32_t dd_bucket(DedupLds& L, const BucketScratch& bk, const DedupArgs& A, uint32_t& parity, HotPlan& P, uint32_t& before) {
    const uint32_t b = dd_unit(A);
    const bool own = b < A.nbk, hash = b < A.nbk_hash;
    const uint32_t tot0 = own ? bk.tot[b] : 0u, tot1 = own ? bk.tot[bk.n_buckets_max + b] : 0u;
    const uint32_t n_hot = A.nbk - A.nbk_hash;
    const uint32_t ht0 = threadIdx.x < n_hot ? bk.tot[A.nbk_hash + threadIdx.x] : 0u, ht1 = threadIdx.x < n_hot ? bk.tot[bk.n_buckets_max + A.nbk_hash + threadIdx.x] : 0u;
    const uint4 hdr = *reinterpret_cast<const uint4*>(bk.seq);
    const DdRuns runs = hash || (own && A.hash_first) ? dd_seg_load(bk, A.nbk, A.part_blocks, A.per_block, b, threadIdx.x) : DdRuns{};
    uint32_t s0 = 0, s1 = 0;   // this thread's share of the totals in front of the bucket (both copies: the selector is not known yet)
    if (PREFIX && hash)
        for (uint32_t j = threadIdx.x; j < b; j += kDedupThreads) { s0 += bk.tot[j]; s1 += bk.tot[bk.n_buckets_max + j]; }
    parity = __builtin_amdgcn_readfirstlane(hdr.y);
    if (b == 0 && threadIdx.x == 0) atomicAdd(&bk.seq[0], 1u);   // this partition is consumed
    P = hot_plan<WIN>(L, bk, A, ht0, ht1, parity, own ? -1 : (int)(b - A.nbk));
    // the pinned host word the next partition sizes itself by: the units this batch has beyond its hash buckets — the hot keys' windows (they keep
    // their buckets while they stay hot) and, added below as they turn up, the slabs an oversized hash bucket would make
    if (b == 0 && threadIdx.x == 0) report_units(bk, A.h_slabs, P.units);
    uint32_t size = __builtin_amdgcn_readfirstlane(parity ? tot1 : tot0);
    before = 0;
    if (!own) {   // a window of a hot key's bucket
        if (!P.valid) return 0u;
        size = P.size;
        dd_seg_scan(L, dd_seg_load(bk, A.nbk, A.part_blocks, A.per_block, A.nbk_hash + P.h, threadIdx.x), threadIdx.x);
    } else {
        if (!hash) {   // a hot key's bucket: its windows' business (the blocks behind the buckets)
            if (A.hash_first && size >= A.hot_count) {   // (block-uniform) ... but for the key's report: it stays listed while it stays hot
                dd_seg_scan(L, runs, threadIdx.x);
                lds_barrier();
                if (threadIdx.x == 0) report_hot_key(bk, parity, bk.ent[dd_entry_at(L, 0)].key, hot_cap_of(size, A.hot_count));
            }
            return 0u;
        }
        dd_seg_scan(L, runs, threadIdx.x);
        if (PREFIX) {
            uint32_t mine = parity ? s1 : s0;
#pragma unroll
            for (int d = 32; d; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d);
            if ((threadIdx.x & 63) == 0) L.pre[threadIdx.x >> 6] = mine;
        }
    }
    lds_barrier();
    if (A.size_from_runs && own) size = __builtin_amdgcn_readfirstlane(L.seg_first[kPartBlocksMax]);   // (dd_seg_scan left the sum of the runs there)
    // a skewed stream: hot keys get buckets of their own next time.  (Only while the plan has none: once it has, the windows alone are what the next partition
    // reserves block slots for — a hash bucket beyond 1024 positions is then just a large bucket, and counting its "slabs" made the next plan's hash buckets fewer
    // and larger still)
    if (own && size > kBucketCap && A.nbk == A.nbk_hash && threadIdx.x == 0) report_slabs(bk, A.h_slabs, (size + kSlab - 1) / kSlab, P.units);
    if (PREFIX && own) for (int w = 0; w < kDedupWaves; ++w) before += L.pre[w];
    lds_barrier();
    return size;
}

__global__ __launch_bounds__(kDedupThreads, 7) void bkt_dedup_keys_kernel(DedupArgs A, BucketScratch bk) {
    __shared__ DedupLds L;
    uint32_t parity, before;
    HotPlan P;
    const uint32_t size = dd_bucket<true>(L, bk, A, parity, P, before);
    if (size == 0) return;
    if (dd_unit(A) >= A.nbk) {   // ---- a window of a hot key's own bucket: every entry is that key, its number is P.rank
        const uint32_t lo = P.win * kHotWindow, hi = min(size, lo + kHotWindow);
        if (P.win == 0 && threadIdx.x == 0) {
            const int64_t key = bk.ent[dd_entry_at(L, 0)].key;
            A.uniq[P.rank] = key;
            if (size >= A.hot_count) report_hot_key(bk, parity, key, hot_cap_of(size, A.hot_count));   // stays listed while it stays hot
        }
        for (uint32_t e0 = lo; e0 < hi; e0 += 4 * kDedupThreads) {   // four positions in flight per thread (a window is 16 per thread: 4 round trips, not 16)
            uint32_t pp[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { const uint32_t e = e0 + q * kDedupThreads + threadIdx.x; pp[q] = e < hi ? bk.ent[dd_entry_at(L, e)].pos : 0xFFFFFFFFu; }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) if (pp[q] != 0xFFFFFFFFu) A.inverse[pp[q]] = (int64_t)P.rank;
        }
        return;
    }
    // Where this bucket's distinct keys go in d_uniq: behind the hot keys' numbers, at the positions the hash buckets in front of it hold —
    // every bucket owns as many entries of the list as it has positions, its distinct keys fill the front of that slice and the rest stays
    // EMPTY.  No counter: 3 000 blocks that reserve their slices from one word take 35 us for that alone (one word serves ~88 atomics per us).
    uint32_t slice = P.H + before;
    dd_passes<false>(L, bk, size, parity, A.hot_count, A.status, [&](uint32_t bits, uint64_t val, const DdHeld& held) {
        const uint32_t t = threadIdx.x;
        // the pass's distinct keys get consecutive numbers (block scan over the table's slots), its slice of the unique list one atomic
        constexpr uint32_t per = kDedupSlots / kDedupThreads;
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) mine += L.key[t * per + q] != 0ull;
        unsigned long long total;
        uint32_t idx = (uint32_t)block_scan_u64<kDedupWaves, true>(mine, L.wsum, total);
        const uint32_t base = slice;
        slice += (uint32_t)total;   // (block-uniform: the next pass of this bucket continues behind this one's keys)
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t s = t * per + q;
            if (L.key[s] != 0ull) { A.uniq[base + idx] = (int64_t)(L.key[s] ^ kBias); L.val[s] = idx++; }
        }
        lds_barrier();   // (the key stores are not read back)
        const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
        if (held.valid) {   // (block-uniform) the entries are still in registers
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (held.s[q] != kNoSlot) A.inverse[held.p[q]] = (int64_t)(base + L.val[held.s[q]]);
            return;
        }
        for (uint32_t e0 = 0; e0 < size; e0 += 4 * kDedupThreads) {   // four entries in flight per thread
            int64_t kq[4]; uint32_t pq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t e = e0 + (uint32_t)q * kDedupThreads + t;
                kq[q] = kEmpty; pq[q] = 0u;
                if (e < size) { const PartEntry en = bk.ent[dd_entry_at(L, e)]; kq[q] = en.key; pq[q] = en.pos; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (kq[q] != kEmpty && (mix64b((uint64_t)kq[q]) & mask) == val) A.inverse[pq[q]] = (int64_t)(base + L.val[dd_lookup(L, (unsigned long long)kq[q] ^ kBias)]);
        }
    });
}

// the partition of a dedup or an assign: the same sort as an apply's, and — while the keys are at hand anyway — the output's padding: d_uniq[i] = EMPTY
// for every i (the dedup kernel overwrites the first n_unique of them), d_inverse[i] = miss_index for reserved keys (they are in no bucket)
__global__ __launch_bounds__(1024) void bkt_sort_dedup_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t nbk_hash, uint32_t nbk, uint32_t per_block, BucketScratch bk,
                                                              uint32_t* status, OpCounters* op, int64_t* __restrict__ uniq, int64_t* __restrict__ inverse, int64_t miss_index,
                                                              uint8_t* __restrict__ found, uint32_t tot_atomics, uint32_t* __restrict__ counts) {
    extern __shared__ unsigned long long part_lds[];
    __shared__ unsigned long long wsum[1024 / 64];
    const uint32_t lo = blockIdx.x * per_block, hi = min(n, lo + per_block);
    if (uniq) {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
            uniq[i] = kEmpty;
            if (counts) counts[i] = 0u;   // (mee_dedup_sum: padding has no occurrences)
            if (inverse && reserved_key(keys[i])) inverse[i] = miss_index;
        }
    } else if (found) {   // assign: every position's found byte starts as 1 (0 for reserved keys: they are in no bucket) — coalesced stores here; the assign
        // kernel then writes only the bytes of keys it does NOT find (1M scattered byte stores cost 10-17 us of a 180 us assign: a partial-line fill each)
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) found[i] = reserved_key(keys[i]) ? 0 : 1;
    }
    PartHot* hot = reinterpret_cast<PartHot*>(part_lds);
    sort_role<1024>(keys, n, nbk_hash, nbk, per_block, blockIdx.x, gridDim.x, bk, status, op, reinterpret_cast<uint32_t*>(hot + 1), wsum, hot, tot_atomics != 0);
}

// ---- assign (update-if-present): last position wins, one probe + one row copy per distinct key ---------------------------------------------
struct AssignArgs {
    DedupArgs d;
    int64_t* tkeys; float4* rows; uint64_t nb; uint32_t dim4;
    const float4* values; uint8_t* found;   // (values: the batch's rows; rows: the table plane)
};

constexpr uint32_t kAssignBlocksPerCU = 5;   // (the register bound below: 90-96 VGPRs)
template <int DIM4>
__global__ __launch_bounds__(kDedupThreads, kAssignBlocksPerCU) void bkt_assign_kernel(AssignArgs A, BucketScratch bk) {
    __shared__ DedupLds L;
    uint32_t parity;
    HotPlan P;
    uint32_t before;
    const uint32_t size = dd_bucket<false>(L, bk, A.d, parity, P, before);
    if (size == 0) return;
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4;
    if (dd_unit(A.d) >= A.d.nbk) {   // ---- a window of a hot key's own bucket: every entry is that key
        const uint32_t t = threadIdx.x, b = A.d.nbk_hash + P.h;
        const int lane = t & 63, tile = lane >> 4, tl = lane & 15;
        const uint32_t lo = P.win * kHotWindow, hi = min(size, lo + kHotWindow);
        const int64_t key = bk.ent[dd_entry_at(L, 0)].key;
        // the window's last position -> the bucket's (atomicMax on the bucket's word; the partition zeroed it)
        uint32_t pm = 0;
        for (uint32_t e0 = lo; e0 < hi; e0 += 4 * kDedupThreads) {   // four positions in flight per thread
            uint32_t pp[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { const uint32_t e = e0 + q * kDedupThreads + t; pp[q] = e < hi ? 1u + bk.ent[dd_entry_at(L, e)].pos : 0u; }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) pm = max(pm, pp[q]);
        }
#pragma unroll
        for (int d = 32; d; d >>= 1) pm = max(pm, (uint32_t)__shfl_xor((int)pm, d));
        if (lane == 0) L.pre[t >> 6] = pm;
        // every window probes for itself (one bucket line): its entries' found bytes need no word from the other windows
        int64_t slot = -1;
        if (t < 64) {
            bool is_new, full;
            slot = tile_locate<false, false>(A.tkeys, A.nb, key, tile == 0, tile, tl, is_new, full);
            if (t == 0) { L.base = (uint32_t)(slot >= 0); L.hp[0] = (uint32_t)slot; L.hp[1] = (uint32_t)((uint64_t)slot >> 32); }
        }
        __syncthreads();
        if (A.found && !L.base) {   // (block-uniform; the partition's launch wrote the 1s)
            for (uint32_t e0 = lo; e0 < hi; e0 += 4 * kDedupThreads) {
                uint32_t pp[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) { const uint32_t e = e0 + q * kDedupThreads + t; pp[q] = e < hi ? bk.ent[dd_entry_at(L, e)].pos : 0xFFFFFFFFu; }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) if (pp[q] != 0xFFFFFFFFu) A.found[pp[q]] = 0;
            }
        }
        if (t == 0) {
            uint32_t m = 0;
            for (int w = 0; w < kDedupWaves; ++w) m = max(m, L.pre[w]);
            (void)__hip_atomic_fetch_max(&bk.pend_cnt[b], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the maximum is in before the ticket is drawn
            const uint32_t tk = __hip_atomic_fetch_add(&bk.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L.overflow = tk == P.n_win - 1;   // the window that finishes last copies the winner's row
            if (tk == P.n_win - 1) L.n_distinct = __hip_atomic_load(&bk.pend_cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u;
            if (P.win == 0 && size >= A.d.hot_count) report_hot_key(bk, parity, key, hot_cap_of(size, A.d.hot_count));   // stays listed while it stays hot
        }
        __syncthreads();
        if (L.overflow && L.base && t < 16) {
            const uint32_t win = L.n_distinct;
            const int64_t sl = (int64_t)((uint64_t)L.hp[0] | (uint64_t)L.hp[1] << 32);
            for (uint32_t col = t; col < dim4; col += 16) A.rows[(uint64_t)sl * dim4 + col] = A.values[(uint64_t)win * dim4 + col];
        }
        return;
    }
    dd_passes<true>(L, bk, size, parity, A.d.hot_count, A.d.status, [&](uint32_t bits, uint64_t val, const DdHeld& held) {
        const uint32_t t = threadIdx.x;
        const int lane = t & 63, tile = lane >> 4, tl = lane & 15, wv = t >> 6;
        // the pass's distinct keys as a dense list (block scan over the table's slots; the occurrence counts have been read: their array holds the list)
        constexpr uint32_t per = kDedupSlots / kDedupThreads;
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) mine += L.key[t * per + q] != 0ull;
        unsigned long long total;
        uint32_t idx = (uint32_t)block_scan_u64<kDedupWaves>(mine, L.wsum, total);
        __syncthreads();   // (every thread has read the counts of its slots in the hot-key report)
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) if (L.key[t * per + q] != 0ull) L.cnt[idx++] = t * per + q;
        __syncthreads();
        const uint32_t n_items = (uint32_t)total;
        // two keys per tile and step: both source rows and both first bucket lines are requested before anything is waited for; a key whose
        // first bucket neither holds it nor ends its probe (2-3 % at load 0.75) goes through the full probe afterwards
        for (uint32_t it0 = (uint32_t)wv * 8; it0 < n_items; it0 += kDedupWaves * 8) {   // wave-uniform bound: ballots below
            uint32_t s[2], win[2];
            int64_t key[2], slot[2];
            bool act[2], pend[2];
            f32x4 row[2];
            int64_t kb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t item = it0 + (uint32_t)r * 4 + tile;
                act[r] = item < n_items;
                s[r] = act[r] ? L.cnt[item] : 0u;
                key[r] = (int64_t)(L.key[s[r]] ^ kBias);
                win[r] = L.val[s[r]] - 1u;
                row[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (act[r] && (uint32_t)tl < dim4) row[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A.values) + (uint64_t)win[r] * dim4 + tl);
                kb[r] = act[r] ? A.tkeys[bucket_of(key[r], A.nb) * kW + tl] : kEmpty;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t tm = tile_bits(__ballot(act[r] && kb[r] == key[r]), tile), te = tile_bits(__ballot(act[r] && kb[r] == kEmpty), tile);
                slot[r] = tm ? (int64_t)(bucket_of(key[r], A.nb) * kW) + (__ffs(tm) - 1) : -1;
                pend[r] = act[r] && !tm && !te;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (__any(pend[r])) {
                    bool is_new, full;
                    const int64_t s2 = tile_locate<false, false>(A.tkeys, A.nb, key[r], pend[r], tile, tl, is_new, full);
                    if (pend[r]) slot[r] = s2;
                }
                if (act[r] && slot[r] >= 0) {
                    f32x4* dst = reinterpret_cast<f32x4*>(A.rows) + (uint64_t)slot[r] * dim4;
                    if ((uint32_t)tl < dim4) dst[tl] = row[r];
                    for (uint32_t col = tl + 16; col < dim4; col += 16) dst[col] = reinterpret_cast<const f32x4*>(A.values)[(uint64_t)win[r] * dim4 + col];
                }
                if (act[r] && tl == 0) L.val[s[r]] = slot[r] >= 0;   // from here on: the key's found flag
            }
        }
        __syncthreads();
        if (A.found && held.valid) {   // (block-uniform) the entries are still in registers
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (held.s[q] != kNoSlot && L.val[held.s[q]] == 0u) A.found[held.p[q]] = 0;   // (the partition's launch wrote the 1s)
        } else if (A.found) {
            const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
            for (uint32_t e = t; e < size; e += kDedupThreads) {
                const uint32_t at = dd_entry_at(L, e);
                const PartEntry en = bk.ent[at];
                const int64_t key = en.key;
                if ((mix64b((uint64_t)key) & mask) != val) continue;
                if (L.val[dd_lookup(L, (unsigned long long)key ^ kBias)] == 0u) A.found[en.pos] = 0;
            }
        }
    });
}

// ---- duplicate-key reduction with row sums (mee_dedup_sum, SPEC.md §4): the rows of a key's occurrences added up in fp64, rounded once -------
// What mee_dedup_keys does, and per distinct key its occurrence count and summed row.  A pass's table gives every run (distinct key) its
// number and — block prefix sums over the table's slots — the place of its sources in a list sorted by run; the sources are filled in through an
// LDS cursor per run; then the runs are summed by as much of the block as their length wants: a run of up to 8 sources by one 16-lane TILE (two
// dependent round trips of four rows), up to 64 by a WAVE (its four tiles, 16 rows per round trip, combined with two shuffles), longer ones by the
// whole BLOCK (64 rows per round trip, the four waves' sums combined through 2 KB of LDS).  The runs are numbered short ones first, so that the
// three classes are three index ranges of one list.  A bucket of more than 1024 positions keeps its sorted sources in global scratch (one
// max_batch-sized array, every bucket owns the range its positions own), written and re-read with agent-scope accesses by the same block.
// A hot key's own bucket (one key) is cut into windows of kSumWindow positions, one block each; a window leaves an fp64 partial row, draws a
// ticket, and the window that draws the last one adds the partial rows up (pending-record discipline of the apply: write-through stores, drained
// before the ticket, agent-scope loads; no fence).
constexpr uint32_t kSumBlocksPerCU = 6, kSumBucketMax = 683;   // (the kernel's register bound; positions per bucket aimed at: see bucket_dedup_sum)
constexpr uint32_t kSumTileMax = 8, kSumWaveMax = 64;   // (rows in flight are registers: 4 per tile everywhere — 8 in the wave- and block-level sums cost 84 B more scratch per lane, paid by every wave)
struct SumLds {
    DedupLds d;
    uint32_t off[kDedupSlots];        // per run: where its sources begin in the sorted list (while the list is filled: the fill cursor)
    uint32_t is_last;
    // Once a pass's keys have been written out and every entry has looked its key up, the key table's 8 KB are free: they hold
    //   src   [kDedupSlots] u32  the sorted list (batch positions), when the unit has at most kDedupSlots positions      (key[0 .. 512))
    //   items [kDedupSlots] u16  run number -> table slot                                                                 (key[512 .. 768))
    //   prow  [kDedupWaves][64] f64  block-level sums: one partial row (64 floats' worth) per wave                        (key[768 .. 1024))
    // — 25 KB less LDS per block than arrays of their own: 7 resident blocks per CU instead of 5.
    __device__ __forceinline__ uint32_t* src() { return reinterpret_cast<uint32_t*>(d.key); }
    __device__ __forceinline__ uint16_t* items() { return reinterpret_cast<uint16_t*>(d.key + 512); }
    __device__ __forceinline__ double* prow(uint32_t w) { return reinterpret_cast<double*>(d.key + 768) + w * 64; }
};
static_assert(kDedupSlots * 8 == kDedupSlots * 4 + kDedupSlots * 2 + kDedupWaves * 64 * 8, "the aliases fill the key table exactly");
struct SumArgs {
    DedupArgs d;
    const float4* grads; float4* gsum; uint32_t* counts; uint32_t dim4;
    uint32_t* src_scratch;            // [max_batch] sorted sources of buckets beyond the LDS list
    double* part; uint32_t max_part;  // fp64 partial rows of the hot keys' windows, one per window unit
    // Hand-over of a hash bucket's LONG runs (skewed batches: the plan has buckets for hot keys, so the grid ends with window blocks, most of them without a window):
    // the bucket's block leaves a long run's sorted sources in the global list and an item (number, place, length) here, kHandK per bucket at most, and publishes
    // 1 + its item count in `hand_flag`; the window blocks — when their own window is done, or at once — take the hash buckets two at a time from `hand_head` (eight at a time left one block with ten runs: 100 us) and
    // sum their items with the whole block, eight rows in flight per tile.  (A key of 900 occurrences that found no number in the hot-key set was 14 dependent round
    // trips of its bucket's block — 40 us on top of the block's 70 — and the kernel ends with its slowest block: `profiles/r05_dedup.md`.)
    uint4* hand_items; uint32_t* hand_flag; uint32_t* hand_head; uint32_t handoff;
};
constexpr uint32_t kHandK = 16, kHandChunk = 2;
struct D4 { double x, y, z, w; };
__device__ __forceinline__ D4 d4_tiles_sum(D4 v) {   // over the wave's four tiles (lanes l, l ^ 16, l ^ 32, l ^ 48)
    v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
    v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    return v;
}
// sources first, first + 1, .. first + 3, then first + step .. of a run of c: four rows in flight (held as fp32: 16 registers, not 32), added up in fp64; a lane past the
// end reads the run's last row again and adds +0.0 (every round trip carries four rows).  load(j, col) = float4 column `col` of the run's source j — f32x4 (a gradient
// row) or D4 (an fp64 partial row).
__device__ __forceinline__ void d4_add(D4& s, const f32x4& g, bool on) { s.x += on ? (double)g.x : 0.0; s.y += on ? (double)g.y : 0.0; s.z += on ? (double)g.z : 0.0; s.w += on ? (double)g.w : 0.0; }
__device__ __forceinline__ void d4_add(D4& s, const D4& g, bool on) { s.x += on ? g.x : 0.0; s.y += on ? g.y : 0.0; s.z += on ? g.z : 0.0; s.w += on ? g.w : 0.0; }
template <uint32_t NF = 4, class Load>   // NF rows in flight; `first` and `step` in sources (a cooperating tile T of nT takes first = NF * T, step = NF * nT)
__device__ __forceinline__ D4 run_sum4(uint32_t first, uint32_t step, uint32_t c, uint32_t col, Load load) {
    D4 s{0.0, 0.0, 0.0, 0.0};
    for (uint32_t q0 = first; q0 < c; q0 += step) {
        decltype(load(0u, 0u)) g[NF];
#pragma unroll
        for (uint32_t q = 0; q < NF; ++q) g[q] = load(min(q0 + q, c - 1), col);
#pragma unroll
        for (uint32_t q = 0; q < NF; ++q) d4_add(s, g[q], q0 + q < c);
    }
    return s;
}
// the whole block sums ONE run of c sources, 16 columns (64 floats) at a time; out(col, total) is called by thread tl < 16 of wave 0 for its column.
// Block-uniform control flow (barriers inside).
template <uint32_t NF = 4, class Load, class Out>
__device__ __forceinline__ void block_run_sum(SumLds& L, uint32_t dim4, uint32_t c, Load load, Out out) {
    const uint32_t t = threadIdx.x, lane = t & 63, tile = lane >> 4, tl = lane & 15, wv = t >> 6, T = wv * 4 + tile;
    for (uint32_t cg = 0; cg < dim4; cg += 16) {
        const uint32_t col = cg + tl;
        D4 v = col < dim4 ? run_sum4<NF>(NF * T, NF * 4 * kDedupWaves, c, col, load) : D4{0.0, 0.0, 0.0, 0.0};
        v = d4_tiles_sum(v);
        if (tile == 0) { double* d = L.prow(wv) + tl * 4; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
        lds_barrier();   // (the rows stored by out() are not read back by this block; a window's partial row is drained before its ticket)
        if (t < 16 && col < dim4) {
            D4 r{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int w = 0; w < kDedupWaves; ++w) { const double* d = L.prow(w) + t * 4; r.x += d[0]; r.y += d[1]; r.z += d[2]; r.w += d[3]; }
            out(col, r);
        }
        lds_barrier();
    }
}
__device__ __forceinline__ f32x4 grad_row4(const float4* __restrict__ grads, uint32_t row, uint32_t dim4, uint32_t col) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grads) + (uint64_t)row * dim4 + col);   // every gradient row is read exactly once
}
__device__ __forceinline__ void store_sum4(float4* __restrict__ gsum, uint32_t u, uint32_t dim4, uint32_t col, const D4& v) {   // (streamed: the rows go to a peer or to an apply, not back into this kernel)
    __builtin_nontemporal_store(f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w}, reinterpret_cast<f32x4*>(gsum) + (uint64_t)u * dim4 + col);
}

template <int DIM4>
__global__ __launch_bounds__(kDedupThreads, kSumBlocksPerCU) void bkt_dedup_sum_kernel(SumArgs A, BucketScratch bk) {   // (6 blocks per CU: 80 registers and a few dwords of scratch; unbounded: 109 and 4 blocks)
    __shared__ SumLds L;
    uint32_t parity, before;
    HotPlan P;
    MEE_STL(0);
    const uint32_t size = dd_bucket<true, kSumWindow>(L.d, bk, A.d, parity, P, before);
    MEE_STL(1);   // the bucket's totals, runs and prefix are in
#if MEE_SUM_TIMELINE
    if (threadIdx.x == 0 && g_sum_dbg && blockIdx.x < 8192) g_sum_dbg[(uint64_t)blockIdx.x * 16 + 15] = (unsigned long long)size | (unsigned long long)(dd_unit(A.d) >= A.d.nbk) << 32;
#endif
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4;
    const uint32_t t = threadIdx.x, lane = t & 63, tile = lane >> 4, tl = lane & 15, wv = t >> 6, T = wv * 4 + tile;
    const uint32_t unit = dd_unit(A.d);
    if (unit >= A.d.nbk) {   // ---- a block behind the buckets: a window of a hot key's own bucket (every entry is that key, its number is P.rank), then the hash buckets' long runs
      auto window = [&]() {
        if (size == 0) return;   // (block-uniform) no window for this block
        const uint32_t x = unit - A.d.nbk, b = A.d.nbk_hash + P.h;
        const uint32_t lo = P.win * kSumWindow, c = min(size, lo + kSumWindow) - lo;
        if (P.win == 0 && t == 0) {
            const int64_t key = bk.ent[dd_entry_at(L.d, 0)].key;
            A.d.uniq[P.rank] = key;
            if (A.counts) A.counts[P.rank] = size;
            if (!A.d.hash_first && size >= A.d.hot_count) report_hot_key(bk, parity, key, hot_cap_of(size, A.d.hot_count));   // stays listed while it stays hot
        }
        {   // the window's positions: four in flight per thread (kSumWindow = 4 x 256)
            uint32_t pp[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { const uint32_t e = q * kDedupThreads + t; pp[q] = e < c ? bk.ent[dd_entry_at(L.d, lo + e)].pos : 0xFFFFFFFFu; }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) if (pp[q] != 0xFFFFFFFFu) { L.src()[q * kDedupThreads + t] = pp[q]; if (A.d.inverse) A.d.inverse[pp[q]] = (int64_t)P.rank; }
        }
        lds_barrier();
        if (!A.grads) return;   // (grid-uniform)
        if (P.n_win > 1 && x >= A.max_part) { if (t == 0) atomicOr(A.d.status, (uint32_t)MEE_STATUS_INTERNAL); return; }   // cannot happen (max_part covers every window a batch can have)
        auto from_list = [&](uint32_t j, uint32_t col) { return grad_row4(A.grads, L.src()[j], dim4, col); };
        // (EIGHT rows in flight per tile here — a window is 64 rows per tile, a chain of 8 round trips instead of 16 — : this branch holds no bucket entries in
        // registers, the kernel's register need is set elsewhere)
        if (P.n_win == 1) { block_run_sum<8>(L, dim4, c, from_list, [&](uint32_t col, const D4& v) { store_sum4(A.gsum, P.rank, dim4, col, v); }); return; }
        block_run_sum<8>(L, dim4, c, from_list, [&](uint32_t col, const D4& v) {
            double* d = A.part + ((uint64_t)x * dim4 + col) * 4;
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(d), (unsigned long long)__double_as_longlong(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(d + 1), (unsigned long long)__double_as_longlong(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(d + 2), (unsigned long long)__double_as_longlong(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(d + 3), (unsigned long long)__double_as_longlong(v.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partial row is out before the ticket is drawn
        __syncthreads();
        if (t == 0) L.is_last = __hip_atomic_fetch_add(&bk.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.n_win - 1;
        __syncthreads();
        if (!L.is_last) return;   // (block-uniform) the window that finishes last adds the bucket's partial rows up
        const uint32_t x0 = x - P.win;
        block_run_sum<4>(L, dim4, P.n_win, [&](uint32_t j, uint32_t col) {
            const double* d = A.part + ((uint64_t)(x0 + j) * dim4 + col) * 4;
            auto ld = [](const double* q) { return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
            return D4{ld(d), ld(d + 1), ld(d + 2), ld(d + 3)};
        }, [&](uint32_t col, const D4& v) { store_sum4(A.gsum, P.rank, dim4, col, v); });
      };
        window();
        MEE_STL(10);   // the block's window is done
        if (!A.handoff) return;   // (grid-uniform)
        for (;;) {   // the hash buckets' long runs, kHandChunk buckets per turn (block-uniform control flow)
            __syncthreads();   // (the window's / the previous item's last readers of the LDS list)
            if (t == 0) L.d.base = __hip_atomic_fetch_add(A.hand_head, kHandChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const uint32_t j0 = L.d.base;
            if (j0 >= A.d.nbk_hash) { MEE_STL(11); return; }   // no bucket left
            if (t < kHandChunk) {   // a bucket's block publishes 1 + its items when its passes are done (every hash bucket's block is dispatched before the first block back here)
                uint32_t f = 1u;
                if (j0 + t < A.d.nbk_hash) {
                    // (bounded: ~30 ms.  The flag always comes — the bucket's block was dispatched before this one and waits for nothing —, but a wave that could
                    // spin for ever is a hung GPU if that reasoning ever fails: then the bucket's long runs stay unsummed and the table's status says so)
                    uint32_t spins = 0;
                    while ((f = __hip_atomic_load(&A.hand_flag[j0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins < (1u << 16)) __builtin_amdgcn_s_sleep(16);
                    if (f == 0u) { atomicOr(A.d.status, (uint32_t)MEE_STATUS_INTERNAL); f = 1u; }
                }
                L.d.hp[t] = f - 1u;
            }
            __syncthreads();
            for (uint32_t jj = 0; jj < kHandChunk; ++jj) {
                const uint32_t ni = L.d.hp[jj];
                for (uint32_t k = 0; k < ni; ++k) {
                    const uint32_t* it = reinterpret_cast<const uint32_t*>(A.hand_items + (uint64_t)(j0 + jj) * kHandK + k);
                    const uint32_t u = __hip_atomic_load(it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), off = __hip_atomic_load(it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   c = __hip_atomic_load(it + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    if (c <= kDedupSlots) {   // (block-uniform) the run's sources into the LDS list first: one round trip for all of them
                        for (uint32_t i = t; i < c; i += kDedupThreads) L.src()[i] = __hip_atomic_load(&A.src_scratch[off + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lds_barrier();
                        block_run_sum<8>(L, dim4, c, [&](uint32_t q, uint32_t col) { return grad_row4(A.grads, L.src()[q], dim4, col); },
                                         [&](uint32_t col, const D4& v) { store_sum4(A.gsum, u, dim4, col, v); });
                    } else if (dim4 * 4 * sizeof(double) <= sizeof(L.d.cnt)) {   // a key with thousands of occurrences and no bucket of its own (yet): the LDS list's worth of
                        // sources at a time, the sums of the pieces added up in LDS (the counters' 4 KB: thread tl < 16 of wave 0 owns its columns)
                        double* acc = reinterpret_cast<double*>(L.d.cnt);
                        for (uint32_t i = t; i < dim4 * 4; i += kDedupThreads) acc[i] = 0.0;
                        for (uint32_t c0 = 0; c0 < c; c0 += kDedupSlots) {
                            const uint32_t cc = min(kDedupSlots, c - c0);
                            __syncthreads();
                            for (uint32_t i = t; i < cc; i += kDedupThreads) L.src()[i] = __hip_atomic_load(&A.src_scratch[off + c0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            lds_barrier();
                            block_run_sum<8>(L, dim4, cc, [&](uint32_t q, uint32_t col) { return grad_row4(A.grads, L.src()[q], dim4, col); },
                                             [&](uint32_t col, const D4& v) { double* a = acc + col * 4; a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w; });
                        }
                        lds_barrier();
                        if (t < 16) for (uint32_t col = t; col < dim4; col += 16) store_sum4(A.gsum, u, dim4, col, D4{acc[col * 4], acc[col * 4 + 1], acc[col * 4 + 2], acc[col * 4 + 3]});
                    } else   // (... and rows of more than 512 floats) every row behind a load of its source
                        block_run_sum<8>(L, dim4, c, [&](uint32_t q, uint32_t col) { return grad_row4(A.grads, __hip_atomic_load(&A.src_scratch[off + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), dim4, col); },
                                         [&](uint32_t col, const D4& v) { store_sum4(A.gsum, u, dim4, col, v); });
                }
            }
        }
    }
    if (unit >= A.d.nbk_hash) return;   // a hot key's bucket: dd_bucket has reported the key
    uint32_t n_handed = 0;   // (block-uniform) long runs this block has handed over
    bool published = false;
    auto publish = [&]() {   // ... and what tells the blocks behind the buckets that its items are complete: drained stores, then the flag (no fence: the items are written through)
        if (!A.handoff || published) return;   // (block-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(&A.hand_flag[unit], 1u + n_handed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = true;
    };
    if (size == 0) { publish(); return; }
    // ---- a hash bucket: its distinct keys go to its own slice of the outputs (as in bkt_dedup_keys_kernel: behind the hot keys' numbers, at the positions the
    // hash buckets in front of it hold), its sorted sources — when the LDS list does not hold them — to the same range of the global source list
    uint32_t slice = P.H + before, src_at = before;
    dd_passes<false>(L.d, bk, size, parity, A.d.hot_count, A.d.status, [&](uint32_t bits, uint64_t val, const DdHeld& held) {
        const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
        MEE_STL(2);   // entries fetched, keys in the LDS table
        // -- 1. run numbers (keys that occur ONCE first, then the other short runs, then medium, then long) and the runs' places in the sorted source list: ONE block scan
        // over the table's slots
        constexpr uint32_t per = kDedupSlots / kDedupThreads;
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t sl = t * per + q, c = L.d.key[sl] != 0ull ? L.d.cnt[sl] : 0u;
            if (c) mine += (c == 1u ? 1ull : c <= kSumTileMax ? 1ull << 10 : c <= kSumWaveMax ? 1ull << 20 : 1ull << 30) | (unsigned long long)c << 40;   // four counts of 10 bits (<= 896 runs) | sources
        }
        unsigned long long total;
        const unsigned long long ex = block_scan_u64<kDedupWaves, true>(mine, L.d.wsum, total);
        const uint32_t n_1 = (uint32_t)total & 0x3FFu, n_f = (uint32_t)(total >> 10) & 0x3FFu, n_m = (uint32_t)(total >> 20) & 0x3FFu, n_l = (uint32_t)(total >> 30) & 0x3FFu, m_src = (uint32_t)(total >> 40);
        const uint32_t n_s = n_1 + n_f;   // short runs: a tile each
        const uint32_t base = slice;
        slice += n_s + n_m + n_l;   // (block-uniform: the next pass of this bucket continues behind this one's keys)
        const bool in_lds = held.valid;   // (block-uniform) a bucket of at most 1024 positions: its entries are in registers, its sorted sources fit the LDS list
        uint32_t* __restrict__ srcg = A.src_scratch + src_at;
        src_at += m_src;
        {
            uint32_t i_1 = (uint32_t)ex & 0x3FFu, i_f = n_1 + ((uint32_t)(ex >> 10) & 0x3FFu), i_m = n_s + ((uint32_t)(ex >> 20) & 0x3FFu), i_l = n_s + n_m + ((uint32_t)(ex >> 30) & 0x3FFu);
            // the sorted source list: the keys that occur once come first, in the order of their numbers — source i of the list IS the position of single i (the copy
            // path reads one LDS word per row) —, the sources of the other runs follow in slot order
            uint32_t o = n_1 + (uint32_t)(ex >> 40) - ((uint32_t)ex & 0x3FFu);
#pragma unroll
            for (uint32_t q = 0; q < per; ++q) {
                const uint32_t sl = t * per + q;
                if (L.d.key[sl] == 0ull) { L.d.cnt[sl] = 0u; continue; }   // (cnt != 0 marks a run from here on: the key table is about to be reused)
                const uint32_t c = L.d.cnt[sl];
                const uint32_t idx = c == 1u ? i_1++ : c <= kSumTileMax ? i_f++ : c <= kSumWaveMax ? i_m++ : i_l++;
                L.d.val[sl] = idx;
                if (c == 1u) L.off[sl] = idx; else { L.off[sl] = o; o += c; }
                A.d.uniq[base + idx] = (int64_t)(L.d.key[sl] ^ kBias);
                if (A.counts) A.counts[base + idx] = c;
            }
        }
        lds_barrier();
        MEE_STL(3);   // scan done, keys and counts written
        // -- 2. every entry of the pass looks its key up: the key's number into d_inverse; a bucket beyond the LDS list files its positions in the global list at once
        uint32_t my_sl[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (in_lds) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                my_sl[q] = held.s[q];
                if (my_sl[q] != kNoSlot && A.d.inverse) A.d.inverse[held.p[q]] = (int64_t)(base + L.d.val[my_sl[q]]);
            }
        } else {
            for (uint32_t e0 = 0; e0 < size; e0 += 4 * kDedupThreads) {   // four entries in flight per thread
                int64_t kq[4]; uint32_t pq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t e = e0 + (uint32_t)q * kDedupThreads + t;
                    kq[q] = kEmpty; pq[q] = 0u;
                    if (e < size) { const PartEntry en = bk.ent[dd_entry_at(L.d, e)]; kq[q] = en.key; pq[q] = en.pos; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (kq[q] == kEmpty || (mix64b((uint64_t)kq[q]) & mask) != val) continue;
                    const uint32_t sl = dd_lookup(L.d, (unsigned long long)kq[q] ^ kBias);
                    if (A.d.inverse) A.d.inverse[pq[q]] = (int64_t)(base + L.d.val[sl]);
                    if (A.grads) __hip_atomic_store(&srcg[atomicAdd(&L.off[sl], 1u)], pq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        MEE_STL(9);   // thread 0's look-ups done, inverse stores issued
        if (!A.grads) return;   // (grid-uniform) keys, counts and inverse only
        if (in_lds) lds_barrier(); else __syncthreads();   // nobody looks a key up any more: the key table's space becomes src / items / prow (beyond the LDS list: the stores of the global list are drained)
        // -- 2b. the LDS list (positions sorted by run, through each run's cursor) and the run list (run number -> slot)
        if (in_lds) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (my_sl[q] != 0xFFFFFFFFu) L.src()[atomicAdd(&L.off[my_sl[q]], 1u)] = held.p[q];
        }
#pragma unroll
        for (uint32_t q = 0; q < per; ++q) { const uint32_t sl = t * per + q; if (L.d.cnt[sl]) L.items()[L.d.val[sl]] = (uint16_t)sl; }
        lds_barrier();
        MEE_STL(4);   // look-ups, inverse, sorted source list
        // from here on run sl's sources are [L.off[sl] - cnt, L.off[sl]) of the list
        auto source = [&](uint32_t at) -> uint32_t { return in_lds ? L.src()[at] : __hip_atomic_load(&srcg[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        // -- 3. long runs: the whole block, one after the other (the block-level sums use prow: all threads pass the barriers inside)
        for (uint32_t j = 0; j < n_l; ++j) {   // block-uniform
            const uint32_t idx = n_s + n_m + j, sl = L.items()[idx], c = L.d.cnt[sl], first = L.off[sl] - c;
            if (A.handoff && n_handed < kHandK) {   // handed over: the run's sources into the global list (there already when the bucket is beyond the LDS list), its item
                if (in_lds) for (uint32_t i = t; i < c; i += kDedupThreads) __hip_atomic_store(&srcg[first + i], L.src()[first + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == 0) {
                    uint32_t* it = reinterpret_cast<uint32_t*>(A.hand_items + (uint64_t)unit * kHandK + n_handed);
                    __hip_atomic_store(it, base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(it + 1, (uint32_t)(srcg - A.src_scratch) + first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(it + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ++n_handed;
                continue;
            }
            block_run_sum(L, dim4, c, [&](uint32_t q, uint32_t col) { return grad_row4(A.grads, source(first + q), dim4, col); },
                          [&](uint32_t col, const D4& v) { store_sum4(A.gsum, base + idx, dim4, col, v); });
        }
        if (L.d.stk_n == 0u) publish();   // (block-uniform) no pass is left on the stack: the items are complete — the blocks behind the buckets need not wait for this block's rows
        // -- 4. medium runs: a wave each
        for (uint32_t j = wv; j < n_m; j += kDedupWaves) {   // wave-uniform
            const uint32_t idx = n_s + j, sl = L.items()[idx], c = L.d.cnt[sl], first = L.off[sl] - c;
            for (uint32_t col = tl; col < dim4 + tl; col += 16) {   // (every lane runs the same number of turns: the shuffles need whole waves)
                D4 v = col < dim4 ? run_sum4<4>(4 * tile, 16, c, col, [&](uint32_t q, uint32_t cc) { return grad_row4(A.grads, source(first + q), dim4, cc); }) : D4{0.0, 0.0, 0.0, 0.0};
                v = d4_tiles_sum(v);
                if (tile == 0 && col < dim4) store_sum4(A.gsum, base + idx, dim4, col, v);
            }
        }
        MEE_STL(5);   // long and medium runs summed
        // -- 5. short runs: a tile each.  Keys that occur ONCE — the bulk of every batch, numbered first — are streamed rows in, rows out, bit for bit (no arithmetic), four
        // in flight per tile and software-pipelined: the NEXT four rows are requested before the current four are stored, so that a step waits for its loads only (the memory
        // counter retires in order: loads issued behind stores would wait for the stores' acknowledgement as well).  Runs of 2-8 sources follow, one run per step (one or two
        // round trips of four rows).  (First form: all short runs in one sequence, a step of four took the copy path only when all four were single — on a Zipf batch three
        // steps of four went the slow way, singles included.)
        constexpr uint32_t kTiles = 4 * kDedupWaves;
        // a step = runs i0, i0 + 16, i0 + 32, i0 + 48 of the singles; a run beyond the last reads row 0 (a valid address: the load stays unconditional)
        auto request = [&](uint32_t i0, f32x4 (&g)[4]) {
            uint32_t row[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t idx = i0 + q * kTiles;
                row[q] = idx < n_1 ? source(idx) : 0u;
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                g[q] = (uint32_t)tl < dim4 ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A.grads) + (uint64_t)row[q] * dim4 + tl) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        if (T < n_1) {
            f32x4 g[4];
            request(T, g);
            for (uint32_t i0 = T; i0 < n_1; i0 += 4 * kTiles) {
                f32x4 gn[4] = {g[0], g[1], g[2], g[3]};
                if (i0 + 4 * kTiles < n_1) request(i0 + 4 * kTiles, gn);
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    if (i0 + q * kTiles < n_1 && (uint32_t)tl < dim4) __builtin_nontemporal_store(g[q], reinterpret_cast<f32x4*>(A.gsum) + (uint64_t)(base + i0 + q * kTiles) * dim4 + tl);
                for (uint32_t col = tl + 16; col < dim4; col += 16) {   // wider rows: the remaining column groups
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        if (i0 + q * kTiles >= n_1) continue;
                        const f32x4 h = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A.grads) + (uint64_t)source(i0 + q * kTiles) * dim4 + col);
                        __builtin_nontemporal_store(h, reinterpret_cast<f32x4*>(A.gsum) + (uint64_t)(base + i0 + q * kTiles) * dim4 + col);
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) g[q] = gn[q];
            }
        }
        for (uint32_t idx = n_1 + T; idx < n_s; idx += kTiles) {
            const uint32_t sl = L.items()[idx], c = L.d.cnt[sl], f0 = L.off[sl] - c;
            for (uint32_t col = tl; col < dim4; col += 16)
                store_sum4(A.gsum, base + idx, dim4, col, run_sum4(0u, 4u, c, col, [&](uint32_t j, uint32_t cc) { return grad_row4(A.grads, source(f0 + j), dim4, cc); }));
        }
#if MEE_SUM_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MEE_STL(6);   // thread 0's short runs done
    });
    publish();
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------
// blocks behind the buckets: one per window a batch of n keys can have in its hot keys' buckets (sum of ceil(size / kHotWindow) <= n / kHotWindow + hot buckets)
static uint32_t hot_window_blocks(const DedupArgs& A, uint32_t n, uint32_t window = kHotWindow) { return A.nbk != A.nbk_hash ? n / window + (A.nbk - A.nbk_hash) : 0u; }
static int dedup_partition(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st, DedupArgs& A, int64_t* d_uniq, int64_t* d_inverse, int64_t miss_index, uint8_t* d_found,
                           uint32_t* d_counts = nullptr, uint32_t slots_of = 0, uint32_t bucket_max_of = 0) {
    uint32_t grid, nbk;
    bool full;
    const uint32_t nbk_hash = bucket_count_for(t, n, st, &grid, &nbk, &full, slots_of, bucket_max_of, &t->bk_dd);
    // (hot keys' buckets whenever the latest batch reported any: a dedup has no FULL / LEAN kernels, its one kernel takes buckets of any size)
    uint32_t blocks, per_block;
    part_geometry(n, 1024, blocks, per_block, kPartBlocks);
    A.nbk = nbk; A.nbk_hash = nbk_hash; A.part_blocks = blocks; A.per_block = per_block; A.hot_count = hot_count_for(n); A.op = t->op; A.h_slabs = t->bk_dd.h_slabs_dev; A.status = &t->ctr->status;
    const bool atom = bucket_totals_by_atomics(blocks, nbk);
    // an assign without hot keys' buckets needs no bucket totals: nobody numbers anything across buckets, and a bucket's size is the sum of its runs
    A.size_from_runs = !atom && !d_uniq && nbk == nbk_hash;
    bkt_sort_dedup_kernel<<<blocks, 1024, sizeof(PartHot) + nbk * 4, st>>>(d_keys, n, nbk_hash, nbk, per_block, t->bk_dd, &t->ctr->status, t->op, d_uniq, d_inverse, miss_index, d_found, atom, d_counts);
    MEE_HIP(hipGetLastError());
    return atom || A.size_from_runs ? MEE_OK : bucket_totals_launch(t, nbk, blocks, st, &t->bk_dd);
}

int bucket_dedup_keys(mee_table* t, const int64_t* d_keys, uint32_t n, int64_t* d_uniq, int64_t* d_inverse, int64_t miss_index, hipStream_t st) {
    DedupArgs A{};
    // (the geometry of mee_dedup_sum — ONE round of six blocks per CU, buckets of up to ~683 positions — instead of the apply's 3072 buckets of 341 in two rounds: a block
    // costs ~16 us of slot time whatever its size; 59.8 -> 55.3 us uniform, 64.7 -> 57.3 us Zipf(1.05) per 1M keys)
    if (int rc = dedup_partition(t, d_keys, n, st, A, d_uniq, d_inverse, miss_index, nullptr, nullptr, t->bk_dd.slots / kApplyBlocksPerCU * kSumBlocksPerCU, kSumBucketMax)) return rc;
    A.uniq = d_uniq; A.inverse = d_inverse;
    bkt_dedup_keys_kernel<<<A.nbk + hot_window_blocks(A, n), kDedupThreads, 0, st>>>(A, t->bk_dd);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// mee_dedup_sum: the distinct keys (padded like mee_dedup_keys), their occurrence counts (0 for padding), the fp64-summed rows of their occurrences (d_grads / d_gsum
// nullable together: keys, counts and inverse only) and each position's index
int bucket_dedup_sum(mee_table* t, const int64_t* d_keys, const float* d_grads, uint32_t n, int64_t* d_uniq, float* d_gsum, uint32_t* d_counts, int64_t* d_inverse, int64_t miss_index,
                     hipStream_t st) {
    SumArgs A{};
    // Geometry of its own: every block pays ~20 us of dependent steps (totals, runs, entries, LDS table, scans, look-ups) before its first row moves, so the rows
    // want FEW, FAT buckets — one round of the kernel's resident blocks (six per CU), up to kSumBucketMax positions on average (Poisson(683) stays 13 sigma below
    // the 1024 entries a block holds in registers and LDS): 1M keys = 1536 buckets of 683 instead of the apply's 3072 of 341 in two rounds.
    if (int rc = dedup_partition(t, d_keys, n, st, A.d, d_uniq, d_inverse, miss_index, nullptr, d_counts, t->bk_dd.slots / kApplyBlocksPerCU * kSumBlocksPerCU, kSumBucketMax)) return rc;
    A.d.uniq = d_uniq; A.d.inverse = d_inverse;
    A.grads = (const float4*)d_grads; A.gsum = (float4*)d_gsum; A.counts = d_counts; A.dim4 = t->dim4;
    A.src_scratch = t->bs.hidx; A.part = t->bk_dd.sum_part; A.max_part = t->bk_dd.sum_part_rows;
    // a key's own bucket is summed by a block per 1024 occurrences, all at once; inside a hash bucket a run of 500 occurrences is 4-8 dependent round trips of ONE
    // block.  So the bar for a bucket of its own is half the apply's here (n / 2048, at least 256): a Zipf(1.05) batch of 1M keys lists ~120 keys
    A.d.hot_count = n / 2048 > kHotCount ? n / 2048 : kHotCount;
    A.d.hash_first = 1u;
    // the long runs' hand-over (SumArgs): the items live in the admission pass's scratch (max_batch x 4 bytes, unused by this operator), the flags and the turn counter
    // in the pending counters of the buckets (zeroed by every partition; the apply's and the assign's, not this kernel's)
    A.hand_items = reinterpret_cast<uint4*>(t->bs.occ); A.hand_flag = t->bk_dd.pend_cnt; A.hand_head = t->bk_dd.pend_cnt + A.d.nbk_hash;
    A.handoff = d_grads && A.d.nbk != A.d.nbk_hash && (uint64_t)A.d.nbk_hash * kHandK * sizeof(uint4) <= (uint64_t)t->max_batch * 4;
    const uint32_t grid = A.d.nbk + hot_window_blocks(A.d, n, kSumWindow);
    if (t->dim4 == 16) bkt_dedup_sum_kernel<16><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    else if (t->dim4 == 32) bkt_dedup_sum_kernel<32><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    else bkt_dedup_sum_kernel<0><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

#if MEE_SUM_TIMELINE
extern "C" int mee_debug_sum_timeline(unsigned long long* host_out, uint64_t n_words) {   // first call arms the buffer, later calls read it
    static unsigned long long* buf = nullptr;
    if (!buf) {
        if (hipMalloc((void**)&buf, 8192 * 16 * 8) != hipSuccess) return 1;
        (void)hipMemset(buf, 0, 8192 * 16 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sum_dbg), &buf, sizeof buf);
        return 0;
    }
    return hipMemcpy(host_out, buf, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

int bucket_assign(mee_table* t, float* plane, const int64_t* d_keys, const float* d_values, uint32_t n, uint8_t* d_found, hipStream_t st) {
    AssignArgs A{};
    // (rounds of the kernel's own resident blocks — five per CU — with buckets of up to ~683 positions: 1M keys = 2560 buckets of ~410 instead of the apply's 3072 of 341;
    // kernel 142.7 -> 140.0 us uniform, 61.5 -> 57.5 us Zipf(1.05) per 1M keys; one round of 1280 buckets of 820: slower; the uniform case is the rows' traffic)
    if (int rc = dedup_partition(t, d_keys, n, st, A.d, nullptr, nullptr, 0, d_found, nullptr, t->bk_dd.slots / kApplyBlocksPerCU * kAssignBlocksPerCU, kSumBucketMax)) return rc;
    A.tkeys = t->keys; A.rows = (float4*)plane; A.nb = t->nb; A.dim4 = t->dim4; A.values = (const float4*)d_values; A.found = d_found;
    const uint32_t grid = A.d.nbk + hot_window_blocks(A.d, n);
    if (t->dim4 == 16) bkt_assign_kernel<16><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    else if (t->dim4 == 32) bkt_assign_kernel<32><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    else bkt_assign_kernel<0><<<grid, kDedupThreads, 0, st>>>(A, t->bk_dd);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

}  // namespace mee

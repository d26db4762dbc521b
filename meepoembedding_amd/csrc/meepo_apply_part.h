// meepo_apply_part.h — the partition half of the bucketed sparse-optimizer apply (see meepo_apply.hip), as device code two kernels share:
// the partition kernel of an apply (meepo_apply.hip) and the training forward mee_find_located_prepare (meepo_table.hip), whose launch gives
// its first blocks this role so that the partition of the step's backward runs beside — hidden behind — the forward's row gather.
#pragma once
#include "meepo_table_int.h"

namespace mee {

constexpr int kPartBlocks = 128;          // blocks that share the partition of one batch, at most (2 x 64: a wave of the apply kernel scans their run lengths, two per lane)
constexpr int kPartBlocksMax = 128;       // ... of a dedup / an election (meepo_dedup.hip).  (256 blocks for 1M-key batches were measured: the partition gets 2.6 us faster, its
                                          // consumers 7 us slower — twice the runs per bucket: DESIGN.md §8)
constexpr uint32_t kSlab = 512;           // positions per slab of a SPLIT bucket (= an apply block's thread count)
constexpr uint32_t kBucketCap = 1024;     // largest bucket ONE apply block takes whole (two positions per thread; its LDS table has this many slots): only
                                          // larger buckets — a key with >= ~700 occurrences in the batch — are split into slabs with pending records and a merge
constexpr uint32_t kBucketMax = 352;      // positions per bucket aimed at, at most (Poisson(352) stays below 512 by 8 sigma: a uniform batch never needs a thread's second position)
constexpr uint32_t kMaxBuckets = 8192;    // the partition keeps one LDS counter per bucket
// Hot keys (a key with >= kHotCount = 256 occurrences in a bucket or in one slab of the latest batch) get a bucket of their OWN in the
// next batch: the apply kernel reports them into a small hash set (BucketScratch::hot_*), the next partition looks every key up in an LDS
// copy of that set and sends a listed key to bucket nbk_hash + its number.  Popular keys stay popular from batch to batch; a wrong guess
// costs time, never results (a listed key that turns out rare is a small bucket of one key).  What it buys: the hash buckets of a skewed
// batch hold no hot key any more — they are not split, their cold keys need no pending records and no merge — and a hot key's own bucket is
// split into slabs of ONE key whose merge adds up one record per slab.
constexpr uint32_t kHotCount = 256;       // occurrences (in one bucket or one slab) that make a key hot: half a slab ...
// ... or n / 1024 of a larger batch: the set numbers the first kHotCap comers, so the bar must leave fewer candidates than that (Zipf(1.05):
// ~68 keys reach 256 occurrences in a batch of 256K, ~250 do in a batch of 1M — but only ~66 reach 1024)
// (a lower bar — 160 or 200 occurrences per 256K keys — lists more keys and gains nothing: Zipf located kernel 56.4 / 55.1 against 54.7-54.9 us)
inline uint32_t hot_count_for(uint64_t n) { const uint64_t c = n / 1024; return c > kHotCount ? (uint32_t)c : kHotCount; }
constexpr uint32_t kHotSlots = 512;       // slots of the hot-key set
constexpr uint32_t kHotCap = 128;         // hot keys that get a bucket (the set takes no more keys once that many are numbered: its load stays ~0.25); a Zipf(1.05) batch of 256K keys lists ~50, one of 1M keys ~120
// waves per SIMD the register allocator must leave room for in the bucket kernel (meepo_apply.hip): 6 = 80 VGPRs, three 512-thread blocks per CU
// (8 = 64 VGPRs, four blocks: measured slower)
constexpr int kApplyWavesPerSimd = 6;
constexpr uint32_t kApplyBlocksPerCU = kApplyWavesPerSimd / 2; // resident blocks of the bucket kernel per CU (512 threads = 2 waves per SIMD each)

// How many buckets a batch of n keys is cut into.  One apply block per bucket, all buckets the same size (a hash), all blocks equally long:
// with B blocks on S resident slots the kernel takes ceil(B / S) rounds, so B is made a MULTIPLE of S = CUs x kApplyBlocksPerCU — 1024
// buckets on 768 slots (the first version: powers of two) ran one full round and then a second with a third of the chip.  Small batches:
// fewer buckets than slots, at least 128 positions each.
inline uint32_t bucket_count_for_host(uint64_t n, uint32_t slots, uint32_t bucket_max = kBucketMax) {
    if (slots == 0) slots = 768;
    if (n <= (uint64_t)slots * 128) return (uint32_t)((n + 127) / 128 ? (n + 127) / 128 : 1);
    uint32_t rounds = (uint32_t)((n + (uint64_t)slots * bucket_max - 1) / ((uint64_t)slots * bucket_max));
    uint32_t nbk = rounds * slots;
    if (nbk > kMaxBuckets) nbk = kMaxBuckets / slots * slots ? kMaxBuckets / slots * slots : kMaxBuckets;
    return nbk;
}
inline void part_geometry(uint32_t n, uint32_t threads, uint32_t& blocks, uint32_t& per_block, uint32_t max_blocks = kPartBlocks) {
    blocks = (n + 4 * threads - 1) / (4 * threads);   // at least 4 keys per thread
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    per_block = (n + blocks - 1) / blocks;
}

// the same bits, scaled the same way, as the key's table bucket (bucket_of): a block's keys live in one contiguous 1/nbk slice of the table
__device__ __forceinline__ uint32_t apply_bucket_of(int64_t key, uint32_t nbk) { return (uint32_t)__umul64hi(mix64((uint64_t)key), (uint64_t)nbk); }
// The pinned host word that tells the next partition how many units the latest batch had beyond its hash buckets (bucket_count_for).  It is written
// only when the value CHANGES (bk.seq[5] shadows it on the device): a uniform stream reports 0 with every batch and need not cross PCIe for that
// (measured: no difference in step time either way, 85.9-88.0 against 87.5-87.7 us per uniform apply; it just keeps the bus quiet).
__device__ __forceinline__ void report_units(const BucketScratch& bk, uint32_t* h_units, uint32_t v) {
    if (bk.seq[5] != v) { bk.seq[5] = v; *h_units = v; }
}
// A block that met a split bucket where the host expected none (LEAN apply kernel, dedup / assign): adds the bucket's slabs to the batch's running sum (seq[4], reset by
// the partition) and publishes the sum.  Several such blocks finish in any order: only a sum larger than everything published for this batch so far (atomicMax on seq[6])
// reaches the shadow word and the host, so a smaller partial sum that lands late never replaces a larger one.
__device__ __forceinline__ void report_slabs(const BucketScratch& bk, uint32_t* h_units, uint32_t mine, uint32_t base) {
    const uint32_t v = base + atomicAdd(&bk.seq[4], mine) + mine;
    if (atomicMax(&bk.seq[6], v) < v) { bk.seq[5] = v; *h_units = v; }
}
// where a key sits in the hot-key set: Fibonacci hashing, ONE multiply — a full mixer here cost the apply kernels (which report hot keys) registers: with
// mix64 the grouped LEAN kernel spilled 12 B to scratch
__device__ __forceinline__ uint32_t hot_slot_of(int64_t key) { return (uint32_t)(((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 40) & (kHotSlots - 1); }
// LDS the partition role needs beside its bucket counters: the copy of the hot-key set
struct PartHot { unsigned long long key[kHotSlots]; uint16_t idx[kHotSlots]; };
// a key's bucket: its own if the key is listed as hot (nbk_total > nbk_hash: hot buckets exist), else by hash
__device__ __forceinline__ uint32_t part_bucket_of(int64_t key, uint32_t nbk_hash, uint32_t nbk_total, const PartHot* hot, uint32_t xcd_split = 0) {
    const uint64_t m = mix64((uint64_t)key);   // ONE mixer per key (its top bits pick the hash bucket); the hot set's slot costs one multiply more (hot_slot_of)
    if (nbk_total != nbk_hash) {
        const unsigned long long bkey = (unsigned long long)key ^ kBias;
        uint32_t h = hot_slot_of(key);
        for (uint32_t tries = 0; tries < kHotSlots; ++tries) {   // (bounded: a set without an empty slot must not trap the probe)
            const unsigned long long k = hot->key[h];
            if (k == bkey) { const uint32_t i = hot->idx[h]; if (i < nbk_total - nbk_hash) return nbk_hash + i; break; }
            if (k == 0ull) break;
            h = (h + 1) & (kHotSlots - 1);
        }
    }
    if (xcd_split != 0 && !(nbk_hash & 1u)) {
        // buckets 2j and 2j + 1 share their pair's hash range unevenly: the apply block of bucket b runs on XCD b % 8, and the read-modify-write stream of the
        // odd XCDs gets ~15 % less of the fabric: xcd_split / 1024 of the range goes to the even bucket
        const uint64_t half = nbk_hash >> 1;
        return 2u * (uint32_t)__umul64hi(m, half) + (uint32_t)(((m * half) >> 54) >= xcd_split);
    }
    return (uint32_t)__umul64hi(m, (uint64_t)nbk_hash);
}

// Inclusive prefix sums over the 64 lanes of a wave on the DPP path (row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then row_bcast:15 into rows
// 1 and 3 and row_bcast:31 into rows 2 and 3; lanes without a source add the `old` operand, 0).  No LDS, no lane-address arithmetic: the
// __shfl_up form computes six (lane - d) << 2 addresses for ds_bpermute, which the compiler hoisted out of the apply kernel's loops and then
// spilled to scratch — a memory round trip in front of every scan of the split role.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
#define MEE_DPP_ADD64(CTRL, ROWS)                                                                                      \
    do {                                                                                                               \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWS, 0xf, false);      \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xf, false); \
        v += (unsigned long long)hi_ << 32 | lo_;                                                                      \
    } while (0)
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v) {
    MEE_DPP_ADD64(0x111, 0xf); MEE_DPP_ADD64(0x112, 0xf); MEE_DPP_ADD64(0x114, 0xf); MEE_DPP_ADD64(0x118, 0xf);
    MEE_DPP_ADD64(0x142, 0xa); MEE_DPP_ADD64(0x143, 0xc);
    return v;
}
#undef MEE_DPP_ADD64

// a barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the global stores in flight (s_waitcnt vmcnt(0)) — stores that nothing in the
// block reads back; their acknowledgement is 2-5 us under load
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix sum of a packed 64-bit value over a block of NW waves; every field of the packed value must stay below its width
// (LDS_ONLY: the barriers inside do not wait for global stores in flight)
template <int NW, bool LDS_ONLY = false>
__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v, unsigned long long* wsum /*[NW]*/, unsigned long long& total) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform: the comparisons below are scalar)
    const unsigned long long incl = wave_incl_scan_u64(v);
    if (LDS_ONLY) lds_barrier(); else __syncthreads();   // wsum may still be read from an earlier call
    if (lane == 63) wsum[w] = incl;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    unsigned long long pre = 0, tot = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) { const unsigned long long x = wsum[ww]; if (ww < w) pre += x; tot += x; }
    total = tot;
    return pre + incl - v;
}

// a key with enough occurrences in this batch to fill half a slab: into the hot-key set the next partition reads (copy `parity`, cleared by this
// batch's partition).  A few hundred calls per skewed batch, none on a uniform one.  The set stops taking keys once kHotCap are numbered (a
// batch with more hot keys than that keeps the first comers): its load stays low and every probe of it ends at an empty slot.
// The set numbers the first comers; when a batch has more candidates than numbers, WHO is left out must not be chance: a key with 80 000 occurrences that
// stays in a hash bucket is one block's work for milliseconds, one with 3 000 for 150 us (the tail of a Zipf batch's dedup_sum: with ONE reserved quarter for keys
// that pass the bar eightfold, 6 keys of 2-4x the bar and 3 of 4-8x were left out of every 1M-key Zipf(1.05) batch — 122 candidates for 128 numbers, 96 of them
// open to all).  So a candidate is admitted while the set holds fewer keys than ITS class may fill (hot_cap_of): the bar itself 88, twice the bar 108, four times
// 120, eight times all 128 — under any Zipf-like stream each class is about as large as all the classes above it together, and the ones left out are the smallest.
__device__ __forceinline__ uint32_t hot_cap_of(uint32_t count, uint32_t bar) {
    return count >= 8 * bar ? kHotCap : count >= 4 * bar ? kHotCap - kHotCap / 16 : count >= 2 * bar ? kHotCap - 5 * kHotCap / 32 : kHotCap - 5 * kHotCap / 16;
}
__device__ __forceinline__ void report_hot_key(const BucketScratch& bk, uint32_t parity, int64_t key, uint32_t cap = kHotCap) {
    if (__hip_atomic_load(&bk.hot_n[parity], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cap) return;
    const unsigned long long bkey = (unsigned long long)key ^ kBias;
    uint32_t h = hot_slot_of(key);   // (the slot part_bucket_of looks at first)
    unsigned long long* set = bk.hot_key + parity * kHotSlots;
    for (uint32_t tries = 0; tries < kHotSlots / 2; ++tries) {
        const unsigned long long old = atomicCAS(&set[h], 0ull, bkey);
        if (old == bkey) return;   // another slab of the key was first
        if (old == 0ull) {
            const uint32_t i = atomicAdd(&bk.hot_n[parity], 1u);
            if (i < kHotCap) bk.hot_idx[parity * kHotSlots + h] = i;   // (read by the NEXT partition: a later kernel)
            return;
        }
        h = (h + 1) & (kHotSlots - 1);
    }
}

// One partition block (block `blk` of `n_blocks`, THREADS threads): sorts ITS share of the batch by bucket, inside its own contiguous slice of
// pos / pkey (LDS histogram -> in-block prefix sum -> LDS cursors).  A block's writes stay inside its slice (16 KB + 32 KB at 4096 keys),
// i.e. in one XCD's L2, where the 4- and 8-byte stores combine into whole lines.  (The first version scattered every key straight to its
// bucket's global position: 256K keys = 512K partial-line stores from 64 CUs on 8 XCDs into the same lines — 19 us; and it needed a count
// kernel in front.)  Per (block, bucket) it leaves the run's length and its start inside the slice: the apply kernel pulls a bucket's
// entries out of the <= 64 slices (short contiguous reads).  It also adds its run lengths to the buckets' totals (one atomic per non-empty
// (block, bucket) pair, spread over nbk words); the add that takes a total beyond what one block holds (kBucketCap) raises `has_split`, which is all the apply kernel's
// apply blocks look at to know whether the batch has slab units.  Nothing is handed from block to block in here: the kernel boundary publishes everything.
// (Two earlier forms: a units kernel of its own behind this one — a dependent launch, 6.5 us for one block's worth of work; the block that
// finished last building the unit list — release fence, ticket, acquire, then the list: the same 6 us at the end of this kernel, and under
// the training forward's row traffic every one of those dependent steps cost microseconds.)
// (Large batches — partition blocks x buckets beyond ~160 000 cells: 410 000 atomics for 1M keys, 25 of the partition's 36 us — skip the atomics:
// bkt_totals_kernel adds the run-length matrix's columns up behind this kernel, 2-3 us.)
// Scratch that the apply kernel of THIS batch reads is reset here for the next but one use: totals and has_split alternate between two
// copies, this launch zeroes the copy the next launch will add to.  Which copy is in use lives on the DEVICE (a captured graph replays the
// same launches, so a host-side toggle would stand still): bk.seq[0] counts partitions that were consumed — the partition reads it, takes
// its low bit as the copy, and leaves that bit in bk.seq[1] for the apply kernel, which reads only seq[1] and bumps only seq[0] (each word
// is written by one kind of kernel and read by the other: the kernel boundary orders them).  `cursor` = nbk words of LDS.
template <int THREADS>
__device__ __forceinline__ void sort_role(const int64_t* __restrict__ keys, uint32_t n, uint32_t nbk_hash, uint32_t nbk /* hash buckets + hot buckets */, uint32_t per_block, uint32_t blk,
                                          uint32_t n_blocks, const BucketScratch& bk, uint32_t* status, OpCounters* op, uint32_t* cursor /*[nbk]*/,
                                          unsigned long long* wsum /*[THREADS / 64]*/, PartHot* hot,
                                          bool tot_atomics = true /* false: the bucket totals are summed up by a small kernel behind this one (bkt_totals_kernel) */,
                                          uint32_t xcd_split = 0 /* part_bucket_of */) {
    // Dependent round trips to memory are what this role costs (beside the training forward's row gather every one of them waits in the same
    // queues as the gather's requests: microseconds each), so it makes two: the keys together with the copy selector, and — at the very end —
    // the returns of the bucket-total atomics, which travel while the entries are scattered.
    constexpr int kKeyGroup = 8;   // keys a thread loads back to back and keeps in registers (a block of THREADS threads takes up to 8 x THREADS keys in one pass)
    for (uint32_t j = threadIdx.x; j < nbk; j += THREADS) cursor[j] = 0u;
    const uint32_t lo = blk * per_block, hi = min(n, lo + per_block);
    const bool in_regs = per_block <= (uint32_t)kKeyGroup * THREADS;   // block-uniform: each thread's keys fit its registers -> ONE pass over the key array
    int64_t kr[kKeyGroup];
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < kKeyGroup; ++q) kr[q] = lo + threadIdx.x + q * THREADS < hi ? keys[lo + threadIdx.x + q * THREADS] : kEmpty;
    }
    const uint32_t parity = bk.seq[0] & 1u;
    // the hot-key set the LATEST apply kernel left (copy parity ^ 1) comes into LDS with the same round trip; the copy this batch's apply kernel
    // will fill (parity) is cleared, shared out over the blocks
    // (BOTH copies are requested, before the selector is known: one round trip instead of two)
    if (nbk != nbk_hash) {
        for (uint32_t j = threadIdx.x; j < kHotSlots; j += THREADS) {
            const unsigned long long k0 = bk.hot_key[j], k1 = bk.hot_key[kHotSlots + j];
            const uint32_t i0 = bk.hot_idx[j], i1 = bk.hot_idx[kHotSlots + j];
            hot->key[j] = parity ? k0 : k1;
            hot->idx[j] = (uint16_t)min(parity ? i0 : i1, 0xFFFFu);
        }
    }
    for (uint32_t j = blk * THREADS + threadIdx.x; j < kHotSlots; j += n_blocks * THREADS) { bk.hot_key[parity * kHotSlots + j] = 0ull; bk.hot_idx[parity * kHotSlots + j] = 0xFFFFFFFFu; }
    if (blk == 0 && threadIdx.x == 0) bk.hot_n[parity] = 0u;
    // housekeeping for the apply kernel of this batch (pending-record counters, slab tickets of every bucket, the partial-row allocator) and
    // for the NEXT partition (the other copy of the totals): shared out over the blocks
    for (uint32_t j = blk * THREADS + threadIdx.x; j < bk.n_buckets_max; j += n_blocks * THREADS) {
        if (j < nbk) { bk.pend_cnt[j] = 0u; bk.ticket[j] = 0u; }
        bk.tot[(parity ^ 1u) * bk.n_buckets_max + j] = 0u;
    }
    if (blk == 0 && threadIdx.x == 0) { op->n_part = 0u; bk.has_split[parity ^ 1u] = 0u; bk.seq[1] = parity; bk.seq[4] = 0u; bk.seq[6] = 0u; }
    __syncthreads();
    bool bad = false;
    // in_regs: the keys' buckets, two per register (< 2^16: kMaxBuckets + kHotCap), for the second pass — in the partition kernel of an apply only:
    // inside the training forward's launch (256-thread blocks, 64 registers for the sake of the find) the second pass looks them up again
    constexpr bool kKeepBuckets = THREADS >= 512;
    uint32_t bid[kKeyGroup / 2];
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < kKeyGroup; ++q) {
            uint32_t bq = 0xFFFFu;
            if (!reserved_key(kr[q])) { bq = part_bucket_of(kr[q], nbk_hash, nbk, hot, xcd_split); atomicAdd(&cursor[bq], 1u); }
            else bad = bad || kr[q] == kReclaimed;   // EMPTY = padding, silent (SPEC.md §2)
            if (kKeepBuckets) bid[q / 2] = q & 1 ? bid[q / 2] | bq << 16 : bq;
        }
    } else {
        for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += kKeyGroup * THREADS) {
            int64_t k[kKeyGroup];
#pragma unroll
            for (int q = 0; q < kKeyGroup; ++q) k[q] = i0 + q * THREADS < hi ? keys[i0 + q * THREADS] : kEmpty;
#pragma unroll
            for (int q = 0; q < kKeyGroup; ++q) {
                if (!reserved_key(k[q])) atomicAdd(&cursor[part_bucket_of(k[q], nbk_hash, nbk, hot, xcd_split)], 1u);
                else bad = bad || k[q] == kReclaimed;
            }
        }
    }
    if (bad && status) atomicOr(status, (uint32_t)MEE_STATUS_RESERVED_KEY);
    __syncthreads();
    const uint32_t per_t = (nbk + THREADS - 1) / THREADS;   // buckets [t * per_t, (t + 1) * per_t) belong to thread t
    unsigned long long sum = 0;
    for (uint32_t q = 0; q < per_t; ++q) {
        const uint32_t b = threadIdx.x * per_t + q;
        if (b < nbk) sum += cursor[b];
    }
    unsigned long long total;
    uint32_t start = (uint32_t)block_scan_u64<THREADS / 64>(sum, wsum, total);
    constexpr uint32_t kTotRegs = THREADS >= 512 ? 2 : 4;   // buckets per thread whose atomic's return is looked at only at the end (batches of up to 1024 buckets: all of them)
    uint32_t tot_before[kTotRegs], tot_add[kTotRegs];
#pragma unroll
    for (uint32_t q = 0; q < kTotRegs; ++q) { tot_before[q] = 0u; tot_add[q] = 0u; }
    auto place = [&](uint32_t q, bool deferred) {
        const uint32_t b = threadIdx.x * per_t + q;
        if (b >= nbk) return;
        const uint32_t c = cursor[b];
        bk.cnt_mat[(uint64_t)blk * nbk + b] = c;
        bk.off_mat[(uint64_t)blk * nbk + b] = start;
        cursor[b] = start;
        start += c;
        if (c && tot_atomics) {
            const uint32_t before = atomicAdd(&bk.tot[parity * bk.n_buckets_max + b], c);
            if (deferred) { tot_before[q < kTotRegs ? q : 0] = before; tot_add[q < kTotRegs ? q : 0] = c; }   // the return travels while the entries are scattered
            else if (before <= kBucketCap && before + c > kBucketCap) bk.has_split[parity] = 1u;   // exactly one add per bucket takes its total beyond what one block holds
        }
    };
#pragma unroll
    for (uint32_t q = 0; q < kTotRegs; ++q) if (q < per_t) place(q, true);
    for (uint32_t q = kTotRegs; q < per_t; ++q) place(q, false);
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < kKeyGroup; ++q) {
            if (reserved_key(kr[q])) continue;
            const uint32_t r = lo + atomicAdd(&cursor[kKeepBuckets ? (bid[q / 2] >> (q & 1 ? 16 : 0)) & 0xFFFFu : part_bucket_of(kr[q], nbk_hash, nbk, hot, xcd_split)], 1u);
            bk.ent[r] = PartEntry{kr[q], lo + threadIdx.x + q * THREADS, 0u};
        }
    } else {
        for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += kKeyGroup * THREADS) {
            int64_t k[kKeyGroup];
#pragma unroll
            for (int q = 0; q < kKeyGroup; ++q) k[q] = i0 + q * THREADS < hi ? keys[i0 + q * THREADS] : kEmpty;
#pragma unroll
            for (int q = 0; q < kKeyGroup; ++q) {
                if (reserved_key(k[q])) continue;
                const uint32_t r = lo + atomicAdd(&cursor[part_bucket_of(k[q], nbk_hash, nbk, hot, xcd_split)], 1u);
                bk.ent[r] = PartEntry{k[q], i0 + q * THREADS, 0u};
            }
        }
    }
    bool crossed = false;
#pragma unroll
    for (uint32_t q = 0; q < kTotRegs; ++q) crossed = crossed || (tot_before[q] <= kBucketCap && tot_before[q] + tot_add[q] > kBucketCap);
    if (crossed) bk.has_split[parity] = 1u;
}

}  // namespace mee

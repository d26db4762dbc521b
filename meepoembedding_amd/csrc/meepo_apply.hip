// meepo_apply.hip — the bucketed sparse-optimizer apply (SPEC.md §4): duplicate-key reduction without a global group table.
//
// Reference anchor: /root/reference/README.md:2 (no code upstream); BASELINE.json north_star: "the sparse-optimizer (Adagrad/Adam)
// scatter-update … wavefront ballot/prefix-sum for duplicate-key reduction".
//
// Rounds 1-2 paid, per key of a batch, one scattered atomic to claim an entry of a global group table, one scattered read of that entry in the main
// pass and one scattered store to release it — ~30 us of a 256K-key batch that has (almost) no duplicates — plus three dependent launches for the
// duplicates (that apply is gone since round 4).  Here the batch is first PARTITIONED by the top bits of mix64(key) into buckets of 128..352 positions
// on average (one small kernel, LDS histograms, no per-key global atomic), and then ONE kernel gives every bucket to one 512-thread block: all
// occurrences of a key are in the same bucket, so the block finds the batch's duplicates in an LDS hash table, sorts the bucket's positions by key with
// LDS prefix sums, and its 32 tiles update each distinct key once — a key that occurs once straight from its gradient row (bit-exact), a key that occurs
// c times from the fp64 sum of its c rows (runs of more than 8 are cut into chunks whose fp64 partial rows the block combines itself).  Two launches
// per apply whatever the key distribution (one behind the training forward, whose launch carries the partition), no per-key global atomic, nothing to
// clean up afterwards.
//
// Skewed streams (round 4).  A bucket of up to 1024 positions is one block's; a larger one is SPLIT into slabs of 512 that go to blocks of their own —
// the AGENTS: the partition of a skewed stream makes as many fewer hash buckets as the latest batch had units beyond them, so that buckets, slabs and the
// hot keys' own buckets together fill one round of the resident block slots and every unit starts with the kernel.  Each slab emits one pending record per
// distinct key (key + fp64 partial sum; write-through stores, a ticket, no fence and no waiting), the slab that draws the last ticket merges the bucket's
// records with the same LDS machinery and applies the updates.  Keys a batch reports as hot get buckets of their own in the next one (meepo_apply_part.h).
// Two kernels: LEAN (block = bucket, no scratch: uniform streams) and FULL (the unit list); see bkt_apply_kernel.
//
// Bucket = top bits of mix64(key) = the bits that pick the key's table bucket (mulhi64): a block's keys live in one contiguous 1/n_buckets
// slice of the table.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "meepo_apply_part.h"

namespace mee {

// Share (per 1024) of a bucket pair's hash range that goes to the even bucket; 0 = even halves (tuning "apply_xcd_split").  Blocks are dealt round-robin over the
// XCDs, and in round 3-4's timelines the read-modify-write stream of the blocks with an odd index ran ~15 % slower (42.8 vs 38.0 us for a block's items; the find shows
// nothing of the kind).  Located LEAN kernel, uniform 256K keys, one box: 0: 63.1-63.4 us, 530: 62.8, 540: 61.7, 550: 60.8, 560: 61.5, 580: 62.4, 600: 64.0 — the wrong
// side costs what the right side gains, and WHICH parity is slow is a property of the device and the driver's placement, not of the architecture.  So it is MEASURED:
// when the first table with an optimizer is created on a device, xcd_probe_kernel runs the apply's access pattern (streamed gradient row, row + state row read, both
// written back, random rows of a 64 MB plane pair) in a grid shaped like the apply's, three times, and the host compares the mean block time by block-index parity:
// the same sign in both measured launches and at least 5 % apart — the even bucket gets the share that equalises the two; anything else — even halves.  ~2 ms, once
// per device and process (mee_device_calibration reports what was measured).
constexpr int kPartThreads = 1024;
constexpr int kApplyThreads = 512;
constexpr int kApplyWaves = kApplyThreads / 64;
constexpr uint32_t kLdsSlots = kBucketCap;   // the LDS hash table: as many slots as a block holds sources (a bucket of kBucketCap DISTINCT keys fills it to the
                                             // last slot — linear probing then costs ~sqrt(n) probes per key, a few us for a bucket no hash produces by chance)
constexpr uint32_t kChunk = 8;            // sources one tile sums in one go (four rows in flight: two dependent round trips); longer runs are cut into chunks
constexpr uint32_t kMaxQuads = 192;       // wave items (quads of four chunks) of one slab: at most m / 32 + (runs longer than a chunk: m / 9) = 32 + 113 at m = 1024
constexpr uint32_t kLdsPartRows = 16;     // fp64 partial rows a slab keeps in LDS (a slab that is ONE key of 512 occurrences: 16 quads)

// Buckets and apply blocks for a batch of n keys.  One block per bucket and one round of equal blocks (bucket_count_for_host) — unless the
// latest batch was skewed: its S slabs were units of their own in front of the buckets, and S + buckets beyond the resident block slots ran as
// a second round behind the first (25 us units: a Zipf(1.05) batch of 256K keys, 146 slabs + 768 buckets on 768 slots, ended at 75-80 us
// instead of ~50).  Key streams keep their skew from batch to batch, so the next batch gets S (+ 1/16) fewer, larger buckets and as many
// blocks as before: slabs and buckets together fill the slots once.  S comes back through a pinned host word the apply kernel writes — read
// here without any synchronisation (a stale or zero value costs time, never results: the kernel works through whatever units there are).
// A launch that is being CAPTURED into a hipGraph is replayed for batches the host never sees: what the host knows about the stream right now is frozen into
// the graph.  Such launches take the FULL kernel (correct and quick for uniform AND skewed batches; LEAN's slow path would be replayed for ever on a stream
// that turns skewed), keep buckets for hot keys (the hot-key set itself lives on the device and follows the stream from replay to replay), and — when the
// host has no skew report at capture time — still leave a twelfth of the block slots to agents.
static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
// (slots_of / bucket_max_of: the geometry of another consumer of the partition — mee_dedup_sum's blocks, six per CU, want ONE round of buckets of up to ~680 positions)
uint32_t bucket_count_for(mee_table* t, uint64_t n, hipStream_t st, uint32_t* grid_out, uint32_t* nbk_total_out, bool* full_out, uint32_t slots_of, uint32_t bucket_max_of,
                          BucketScratch* state) {
    BucketScratch& sk = state ? *state : t->bk;
    const uint32_t slots = slots_of ? slots_of : sk.slots;
    uint32_t full = bucket_count_for_host(n, slots, bucket_max_of ? bucket_max_of : sk.bucket_max ? sk.bucket_max : kBucketMax);
    // The scratch (totals, run matrices, pending counters, tickets) is strided for n_buckets_max buckets, sized at creation for the DEFAULT bucket size at
    // max_batch (+ the hot keys' buckets): a smaller "apply_bucket_max" must not ask for more buckets than that — it gets larger buckets instead.
    const uint32_t room = sk.n_buckets_max - kHotCap;
    if (full > room) full = room >= slots ? room / slots * slots : room;
    uint32_t nbk = full;
    const bool capturing = sk.skew_adapt && stream_is_capturing(st);
    const uint32_t s_prev = sk.h_slabs && sk.skew_adapt ? *(volatile uint32_t*)sk.h_slabs : 0u;
    // which kernel (bkt_apply_kernel): FULL behind a skewed batch — and for the 64 batches after the last one: a stream whose skew comes and
    // goes must not fall into the LEAN kernel's slow path every other batch —, else LEAN
    if (s_prev) sk.skew_sticky = 64;
    else if (sk.skew_sticky) --sk.skew_sticky;
    if (full_out) *full_out = sk.kernel_choice >= 0 ? sk.kernel_choice != 0 : (capturing || s_prev != 0 || sk.skew_sticky != 0);
    // (a batch with more keys than its buckets hold whole: every bucket is a list of slabs — the FULL kernel's business, whatever the knob says)
    if (full_out && n > (uint64_t)full * (kBucketCap * 3 / 4)) *full_out = true;
    const uint32_t units = s_prev ? s_prev : capturing && full_out && *full_out ? slots / 12 : 0u;
    if (units && n > (uint64_t)slots * 128) {
        uint32_t adj = units + units / 16 + 1;
        if (adj > slots / 2) adj = slots / 2;
        if (adj > full / 2) adj = full / 2;
        nbk = full - adj;
        if (!bucket_max_of) while ((uint64_t)nbk * 2 * kBucketMax < n && nbk < full) ++nbk;   // (never more than ~700 positions per bucket on average: kBucketCap stays 12 sigma away)
        // (a consumer that asked for its own bucket size — mee_dedup_sum, whose units beyond the hash buckets are windows of 1024 positions of the hot keys' buckets,
        // about three quarters full on average: what is left for the hash buckets must still fit them.  The block slots the windows get this way run them beside the
        // hash buckets from the kernel's first microsecond instead of behind them)
        if (bucket_max_of) {
            const uint64_t in_windows = (uint64_t)units * 768, rest = n > in_windows ? n - in_windows : 0;
            while ((uint64_t)nbk * bucket_max_of < rest && nbk < full) ++nbk;
        }
    }
    if (grid_out) *grid_out = full;
    // behind a skewed batch the keys that batch reported as hot get buckets of their own, behind the hash buckets (meepo_apply_part.h) — the
    // FULL kernel's business
    const bool hot = (s_prev || capturing) && nbk + kHotCap <= sk.n_buckets_max && (!full_out || *full_out);
    if (nbk_total_out) *nbk_total_out = hot ? nbk + kHotCap : nbk;
    return nbk;
}

// ---- the partition kernel of an apply (the role itself: meepo_apply_part.h) ----------------------------------------------------------------
__global__ __launch_bounds__(kPartThreads) void bkt_sort_kernel(const int64_t* __restrict__ keys, uint32_t n, uint32_t nbk_hash, uint32_t nbk,
                                                                uint32_t per_block, BucketScratch bk, uint32_t* status, OpCounters* op, uint32_t tot_atomics, uint32_t xcd_split) {
    extern __shared__ unsigned long long part_lds[];   // PartHot, then one counter per bucket
    __shared__ unsigned long long wsum[kPartThreads / 64];
    PartHot* hot = reinterpret_cast<PartHot*>(part_lds);
    sort_role<kPartThreads>(keys, n, nbk_hash, nbk, per_block, blockIdx.x, gridDim.x, bk, status, op, reinterpret_cast<uint32_t*>(hot + 1), wsum, hot, tot_atomics != 0, xcd_split);
}

// the bucket totals of a large batch: column sums of the run-length matrix (one thread per bucket, coalesced along the buckets), and the
// "some bucket is split" flag — what the partition blocks' atomics do for batches of up to ~512K keys
__global__ __launch_bounds__(256) void bkt_totals_kernel(BucketScratch bk, uint32_t nbk, uint32_t part_blocks) {
    // 64 buckets per block; wave w adds up the rows w, w + 4, … of its 64 columns, eight loads in flight; LDS combines the four partial sums
    __shared__ uint32_t part[4][64];
    const uint32_t b = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    uint32_t sum = 0;
    if (b < nbk) {
        for (uint32_t k0 = w; k0 < part_blocks; k0 += 32) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) { const uint32_t k = k0 + 4 * q; v[q] = k < part_blocks ? bk.cnt_mat[(uint64_t)k * nbk + b] : 0u; }
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) sum += v[q];
        }
    }
    part[w][threadIdx.x & 63] = sum;
    __syncthreads();
    if (w == 0 && b < nbk) {
        const uint32_t parity = bk.seq[0] & 1u;   // (the partition in front of this kernel filled this copy; nothing has consumed it yet)
        const uint32_t tot = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        bk.tot[parity * bk.n_buckets_max + b] = tot;
        if (tot > kBucketCap) bk.has_split[parity] = 1u;
    }
}
bool bucket_totals_by_atomics(uint32_t blocks, uint32_t nbk) { return (uint64_t)blocks * nbk <= 160000; }
int bucket_totals_launch(mee_table* t, uint32_t nbk, uint32_t blocks, hipStream_t st, const BucketScratch* bk) {
    bkt_totals_kernel<<<(nbk + 63) / 64, 256, 0, st>>>(bk ? *bk : t->bk, nbk, blocks);
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

// ---- the apply kernel -------------------------------------------------------------------------------------------------------------
struct ApplyLds {
    unsigned long long key[kLdsSlots];   // key ^ kBias, 0 = empty
    long long slot[kLdsSlots];           // per run: the table slot its handle names (LOCATED kernels)
    uint32_t cnt[kLdsSlots];             // occurrences of the run's key in this slab
    uint32_t off[kLdsSlots];             // where the run starts in src
    uint16_t run[kLdsSlots];             // slabs of split buckets: the run's pending-record number inside the slab
    uint32_t src[kBucketCap];            // the slab's sources sorted by run: gradient-row index (positions) | pending-record index (merge)
    uint16_t items[kBucketCap];          // tile items: the slots of the runs of up to kChunk sources (one tile sums and finishes such a run)
    uint32_t quad[kMaxQuads];            // wave items: run slot | quad of the run << 10 | fp64 partial row << 16 (a quad = 4 chunks of kChunk sources)
    uint32_t big[32];                    // runs of more than one quad: run slot | first partial row << 10 | quads << 22
    unsigned long long wsum[kApplyWaves];
    unsigned long long stk_val[72];      // merge: hash prefixes still to do
    uint32_t stk_bits[72];
    unsigned long long kmin, kmax;       // merge: smallest / largest biased key among a pass's candidates (equal: the pass holds ONE key)
    uint32_t seg_first[kPartBlocks + 1]; // where partition block k's entries of this bucket begin within the bucket (prefix of the run lengths)
    uint32_t seg_at[kPartBlocks];        // ... and where that run lies in pos / pkey
    uint32_t n_items, n_quads, n_big, n_runs, part_base, rec_base, is_last, n_cand, stk_n, next_turn;
    uint32_t u_b, u_sub, u_size, u_beg;   // skewed batches: the slab the block works on
    uint32_t pre_slabs[kApplyThreads + 1], pre_pos[kApplyThreads + 1], pre_a[kApplyThreads];   // (pre_a: the total of the thread's first bucket)
    alignas(16) double prow[kLdsPartRows][64];   // fp64 partial rows of the slab's long runs when there are few and the rows are short (dim <= 64): no trip through memory
    uint32_t part_lds;   // skewed batches: slabs / positions in front of each thread's buckets
#if MEE_APPLY_TIMELINE
    uint32_t tl_w;
#endif
};

// MEE_APPLY_TIMELINE (diagnostic builds only: tools/apply_timeline.py): thread 0 of every block stamps the 100 MHz wall clock at the phase
// boundaries of its bucket into a buffer the host reads back (mee_debug_timeline).
#ifndef MEE_APPLY_TIMELINE
#define MEE_APPLY_TIMELINE 0
#endif
#if MEE_APPLY_TIMELINE
#define MEE_TL(A_, i) do { if (threadIdx.x == 0) (A_).dbg[(uint64_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
// blocks of a skewed batch (skew_units): 128 words per block behind that area — [0] entry, [1] units taken, [2] scan done, then 16 words per slab unit taken
#define MEE_TLS(A_, e_, w_, v_) do { if (threadIdx.x == 0 && (w_) < 128) (A_).dbg[16384ull * 8 + (uint64_t)(e_) * 128 + (w_)] = (v_); } while (0)
// ... of which process_slab stamps its own phases: words 8..11 of the slab pass, 12..15 of the latest merge pass (L.tl_w = the slab's first word)
#define MEE_TLP(A_, L_, k_) do { if (SPLIT && threadIdx.x == 0) (A_).dbg[16384ull * 8 + (L_).tl_w + (src_rec ? 12 : 8) + (k_)] = wall_clock64(); } while (0)
#else
#define MEE_TL(A_, i) do { } while (0)
#define MEE_TLS(A_, e_, w_, v_) do { } while (0)
#define MEE_TLP(A_, L_, k_) do { } while (0)
#endif

// how an update's rows are stored: plain stores (streaming and write-through stores measured: nothing moves)
__device__ __forceinline__ void store_row4(float4* p, const float4& v) { *p = v; }

struct ApplyArgs {
    const int64_t* tkeys; float4 *values, *s1, *s2; uint64_t nb; uint32_t dim4;
    const float4* grads; const uint32_t* gidx; const int64_t* slots;
    uint64_t capacity; int64_t handle_tag; uint32_t* status;
    double* part; uint32_t max_part;     // fp64 partial rows of long runs (BatchScratch::gacc)
    uint32_t nbk, part_blocks, per_block;   // the partition: buckets (hash buckets, then one per hot key), partition blocks, batch positions per partition block
    uint32_t nbk_hash;                      // ... the hash buckets among them
    uint32_t hot_count;                     // occurrences in one bucket or slab that make a key hot (its own bucket in the next batch)
    uint32_t* h_slabs;                      // pinned host word: the slabs this batch's split buckets were cut into (0: none)
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

// Pending records (key, slot, fp64 partial row) go from the slab that writes them to the block that merges the bucket — another CU, most
// likely another XCD with its own L2.  They are written with agent-scope (sc1: write-through) stores and read with agent-scope loads; the
// producer drains its stores (s_waitcnt vmcnt(0)) before it draws its ticket.  No release / acquire fence: an agent-scope release is a
// write-back of the XCD's whole L2, full of the rows the batch has just updated — MI355X_MICROARCH.md "publish-large": 8.2 us against 3.0
// for a 64 KB slab, and the hand-off of a slab measured 6-7 us with the fence pair (tools/apply_timeline.py).
__device__ __forceinline__ void rec_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_store(int64_t* p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double rec_load(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int64_t rec_load(const int64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rec_store_row4(double* dst, double x, double y, double z, double w) { rec_store(dst, x); rec_store(dst + 1, y); rec_store(dst + 2, z); rec_store(dst + 3, w); }

// sum over the wave's four tiles (lane l, l ^ 16, l ^ 32, l ^ 48 hold the same column of four different partial sums): every lane gets the total
__device__ __forceinline__ double tiles_sum(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// nh rows of A.grads (fp32) | of the pending records (fp64), four in flight, added up in fp64; a lane past the end reads the last row again and
// adds +0.0 (no one-row-at-a-time tail: every round trip carries four rows)
template <bool REC>
__device__ __forceinline__ void sum_sources(const ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, uint32_t rec_bucket0, uint32_t run0, uint32_t nh,
                                            uint32_t dim4, uint32_t col, double& sx, double& sy, double& sz, double& sw) {
    if constexpr (REC) {
        for (uint32_t q0 = 0; q0 < nh; q0 += 4) {
            double2 lo[4], hi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* r = bk.pend_row + ((uint64_t)(rec_bucket0 + L.src[run0 + min(q0 + q, nh - 1)]) * dim4 + col) * 4;
                lo[q] = make_double2(rec_load(r), rec_load(r + 1)); hi[q] = make_double2(rec_load(r + 2), rec_load(r + 3));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = q0 + q < nh;
                sx += on ? lo[q].x : 0.0; sy += on ? lo[q].y : 0.0; sz += on ? hi[q].x : 0.0; sw += on ? hi[q].y : 0.0;
            }
        }
    } else {
        for (uint32_t q0 = 0; q0 < nh; q0 += 4) {
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
}

// One slab: m sources -> one update (or one pending record) per distinct key.  m <= kBucketCap in the kernel of the whole buckets (SPLIT =
// false: a thread takes sources t and t + 512), m <= kSlab in the split role.
//   src_rec = false: the sources are entries [first, first + m) of bucket b (batch positions + keys, pulled out of the partition blocks' slices
//                    of bk.pos / bk.pkey); a source's row is a row of grads.
//   src_rec = true : (merge of a split bucket) the sources are the pending records whose bucket-relative numbers lie in L.src[0 .. m); a
//                    source's row is the record's fp64 partial sum.
//   emit: results become pending records of bucket `b` instead of table updates (slab of a split bucket).
// Work items.  kChunk = 8 sources is what one TILE sums in one go (four rows in flight: two dependent round trips).  A run of up to 8 sources
// is a tile item: one tile sums it and finishes it.  A longer run is cut into chunks of 8 and handed out as QUADS of four chunks: the four
// tiles of a wave sum one chunk each and add their sums up with two shuffles — a run of up to 32 sources is finished right there, a longer
// one leaves one fp64 partial row per quad (in LDS when the slab has at most 16 of them and dim <= 64), and after a barrier a wave adds the
// rows (tile u: rows u, u + 4, …) and finishes the run.  A slab that is ONE hot key (512 occurrences) is 16 quads on 8 waves, then 16 partial
// rows on one wave: 2 x 2 + 1 dependent round trips.  (Round 3: chunks of 32 as tile items, one partial row per chunk, ONE tile adding them
// two at a time: 8 + 8 round trips, the slab pass of a split bucket 30-45 us.  Chunks of ceil(m / 32) sources — "as many items as tiles" —:
// a bucket of 800 positions around a key of 500 occurrences spent 7 round trips in each of that key's quads.)  Waves take turns from an LDS
// counter, quads first (a quad is up to 32 rows, a key that occurs once is one).

// MODE: kWhole — a whole bucket (sources: batch positions, results: table updates); kEmit — a slab of a split bucket (sources: batch positions,
// results: pending records); kMerge — a merge pass (sources: pending records, results: table updates).  Compile-time: each instance carries only
// what its mode needs (a slab never touches a table row, a merge never reads a gradient row), which is what keeps the skewed path's register
// need near the whole buckets' 77 — every byte of scratch the kernel declares costs the UNIFORM batches time although they never touch it
// (measured: 0 B: 62.5 us, 56 B (round 3): 63.1, 128-160 B: 67-68, 600 B: 94 us for the same bucket path).
// kListed — kWhole over the m entries of bucket b whose indices the caller left in L.off[0 .. m) (a split bucket without its majority key: slow_bucket).
constexpr int kWhole = 0, kEmit = 1, kMerge = 2, kListed = 3;
template <int KIND, int DIM4, bool LOCATED, int MODE, bool GROUPED = false>
__device__ __forceinline__ void process_slab(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, uint32_t first, uint32_t m,
                                             uint32_t b, uint32_t rec_bucket0 /* split buckets: the bucket's first pending record */, uint32_t parity_rt, const GroupDesc* gdesc = nullptr) {
    constexpr bool SPLIT = MODE == kEmit || MODE == kMerge, emit = MODE == kEmit, src_rec = MODE == kMerge, listed = MODE == kListed;
    static_assert(!GROUPED || LOCATED, "a group's batch names rows, not keys");
    constexpr int PPT = SPLIT ? 1 : (int)(kBucketCap / kApplyThreads);   // sources per thread
    // The kernel calls this from a loop (slab, then merge passes).  The thread index is re-read through an empty asm in every call so that
    // nothing derived from it looks loop-invariant: hoisted out of that loop, the per-thread address arithmetic of every array touched in
    // here stayed live across the whole kernel (110 VGPRs instead of 64: half the resident blocks per CU for every block's hot path).
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4;
    constexpr uint32_t lc = kChunk;
    OptArgs a = A.a;
    a.kind = KIND;
    // ---- 1. the slab's keys into the LDS hash table: run = LDS slot, r = arrival number inside the run ----
    uint32_t my_src[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) my_src[u] = 0u;
    if (src_rec) my_src[0] = t < m ? L.src[t] : 0u;   // read before the table is cleared / src is rewritten
    __syncthreads();   // (src_rec = false: the caller's wave 0 has just written L.seg_first / L.seg_at)
    for (uint32_t j = t; j < kLdsSlots; j += kApplyThreads) { L.key[j] = 0ull; L.cnt[j] = 0u; }
    if (t == 0) { L.n_big = 0u; L.next_turn = 0u; }
    __syncthreads();
    uint32_t my_sr[PPT];   // the source's run (LDS slot, 10 bits) | its arrival number inside the run << 10 | first of its run << 31
    int64_t my_tslot[PPT], my_key[PPT];
    // (the global loads of all of a thread's sources are requested before the first LDS insertion waits for any of them)
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const uint32_t ti = t + (uint32_t)u * kApplyThreads;
        my_key[u] = kEmpty; my_tslot[u] = -1; my_sr[u] = 0u;
        if (ti < m) {
            if (src_rec) {
                my_key[u] = rec_load(bk.pend_key + rec_bucket0 + my_src[u]);
                if constexpr (LOCATED) my_tslot[u] = rec_load(bk.pend_slot + rec_bucket0 + my_src[u]);   // the table slot the record carries
            } else {
                const uint32_t gi = listed ? L.off[ti] : first + ti;   // index within the bucket (listed: off is rewritten behind the barrier that ends this phase) -> its run (binary search over the run starts) -> its place in pos / pkey
                uint32_t k = 0;
#pragma unroll
                for (uint32_t stp = kPartBlocks / 2; stp; stp >>= 1) if (L.seg_first[k + stp] <= gi) k += stp;
                const uint32_t at = L.seg_at[k] + (gi - L.seg_first[k]);
                const PartEntry en = bk.ent[at];
                const uint32_t p = en.pos;
                my_key[u] = en.key;
                my_src[u] = A.gidx ? min(A.gidx[p], a.grad_rows - 1) : p;   // the row of the grad array that belongs to the position
                if constexpr (LOCATED) my_tslot[u] = A.slots[p];   // the raw handle: decoded where it is first needed, so that the load travels beside the LDS work
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        if (t + (uint32_t)u * kApplyThreads < m) {
            const int64_t key = my_key[u];
            const unsigned long long bkey = (unsigned long long)key ^ kBias;
            uint32_t sl = (uint32_t)(mix64b((uint64_t)key) & (kLdsSlots - 1));   // mix64's top bits chose the bucket: take another mixer here
            bool first_of_run = false;
            while (true) {
                const unsigned long long old = atomicCAS(&L.key[sl], 0ull, bkey);
                if (old == 0ull) { first_of_run = true; break; }
                if (old == bkey) break;
                sl = (sl + 1) & (kLdsSlots - 1);
            }
            my_sr[u] = sl | atomicAdd(&L.cnt[sl], 1u) << 10 | (uint32_t)first_of_run << 31;
        }
    }
    __syncthreads();
    if constexpr (!SPLIT) MEE_TL(A, 2);   // entries fetched, keys in the LDS table
    MEE_TLP(A, L, 0);
    // ---- 2. prefix sums over the runs: where each run starts in src, its work items, its number, its fp64 partial rows ----
    {
        const uint32_t s0 = 2 * t, c0 = L.cnt[s0], c1 = L.cnt[s0 + 1];
        const uint32_t t0 = c0 != 0 && c0 <= lc, t1 = c1 != 0 && c1 <= lc;                                        // tile items
        const uint32_t q0 = c0 > lc ? ((c0 + lc - 1) / lc + 3) / 4 : 0u, q1 = c1 > lc ? ((c1 + lc - 1) / lc + 3) / 4 : 0u;   // quads
        const uint32_t p0 = q0 > 1 ? q0 : 0u, p1 = q1 > 1 ? q1 : 0u;                                                // partial rows: one per quad of a run of several
        // five fields of 12 bits: sources | tile items | runs | quads | partial rows — each at most kBucketCap in total
        const unsigned long long packed = (unsigned long long)(c0 + c1) | (unsigned long long)(t0 + t1) << 12 | (unsigned long long)((c0 != 0) + (c1 != 0)) << 24 |
                                          (unsigned long long)(q0 + q1) << 36 | (unsigned long long)(p0 + p1) << 48;
        unsigned long long total;
        const unsigned long long ex = block_scan_u64<kApplyWaves>(packed, L.wsum, total);
        uint32_t so = (uint32_t)ex & 0xFFFu, io = (uint32_t)(ex >> 12) & 0xFFFu, ro = (uint32_t)(ex >> 24) & 0xFFFu, qo = (uint32_t)(ex >> 36) & 0xFFFu, po = (uint32_t)(ex >> 48) & 0xFFFu;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t s = s0 + h, c = h ? c1 : c0, nq = h ? q1 : q0;
            L.off[s] = so;
            if (c) {
                L.run[s] = ro;
                if (!src_rec && c >= A.hot_count) report_hot_key(bk, parity_rt, (int64_t)(L.key[s] ^ kBias));   // a key that fills half a slab: its own bucket next time
                if (c <= lc) L.items[io++] = s;
#pragma unroll 1
                for (uint32_t j = 0; j < nq; ++j) if (qo + j < kMaxQuads) L.quad[qo + j] = s | j << 10 | (nq > 1 ? po + j : 0u) << 16;
                if (nq > 1) { L.big[atomicAdd(&L.n_big, 1u) & 31u] = s | po << 10 | nq << 22; po += nq; }   // fewer than 32 such runs: m / 33
                so += c; qo += nq; ++ro;
            }
        }
        if (t == 0) {
            L.n_items = (uint32_t)(total >> 12) & 0xFFFu;
            L.n_runs = (uint32_t)(total >> 24) & 0xFFFu;
            L.n_quads = min((uint32_t)(total >> 36) & 0xFFFu, kMaxQuads);
            const uint32_t np = (uint32_t)(total >> 48) & 0xFFFu;
            const bool in_lds = np <= kLdsPartRows && dim4 <= 16;
            L.part_lds = in_lds;
            uint32_t pb = np && !in_lds ? atomicAdd(&A.op->n_part, np) : 0u;
            if (pb + np > A.max_part) { pb = 0u; atomicOr(A.status, (uint32_t)MEE_STATUS_INTERNAL); }   // cannot happen (max_part covers n / 8 rows: one per 32 sources of a run longer than 32); never write out of bounds
            L.part_base = pb;
            if (emit) L.rec_base = atomicAdd(&bk.pend_cnt[b], L.n_runs);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        if (t + (uint32_t)u * kApplyThreads < m) {
            const uint32_t my_slot = my_sr[u] & 1023u;
            L.src[L.off[my_slot] + ((my_sr[u] >> 10) & 0x1FFFFFu)] = my_src[u];
            if constexpr (LOCATED) {
                if (my_sr[u] >> 31) {   // every occurrence of a key names the same slot: the run keeps one
                    if (src_rec) L.slot[my_slot] = my_tslot[u];
                    else if constexpr (GROUPED) L.slot[my_slot] = my_tslot[u] >= 0 && ((uint64_t)my_tslot[u] >> kGroupSlotBits) < A.n_tables ? my_tslot[u] : -1;
                    else {
                        bool stale;
                        L.slot[my_slot] = handle_slot(my_tslot[u], A.handle_tag, A.capacity, stale);
                        if (stale) atomicOr(A.status, (uint32_t)MEE_STATUS_STALE_HANDLE);   // rare: the caller kept handles across a remove / clear / reserve
                    }
                }
            }
        }
    }
    __syncthreads();
    if constexpr (!SPLIT) MEE_TL(A, 3);   // scans done, sources sorted (LOCATED: slot handles arrived)
    MEE_TLP(A, L, 1);
    const uint32_t n_items = L.n_items, n_quads = L.n_quads, n_big = L.n_big;
    const bool part_lds = L.part_lds != 0;
    const uint32_t n_turns = n_quads + (n_items + 3) / 4;
    const uint32_t rec_out0 = rec_bucket0 + (emit ? L.rec_base : 0u);   // emit: where this slab's records go
    // ---- 3. work items: a wave per turn — a quad (its four tiles sum one chunk each of ONE run) or four tile items ----
    [[maybe_unused]] uint32_t tl_turn = 0;
    // (a wave asks for its NEXT turn before it works on the current one: the counter's answer travels while the rows do)
    uint32_t turn_next = 0;
    if (lane == 0) turn_next = atomicAdd(&L.next_turn, 1u);
    while (true) {
        const uint32_t turn = __builtin_amdgcn_readfirstlane(turn_next);
        if (turn < n_turns && lane == 0) turn_next = atomicAdd(&L.next_turn, 1u);
#if MEE_APPLY_TIMELINE
        if (!SPLIT && threadIdx.x == 0 && tl_turn < 16) A.dbg[16384ull * 8 + (uint64_t)blockIdx.x * 128 + 68 + 2 * tl_turn] = wall_clock64() << 8 | (turn >= n_turns ? 3u : turn < n_quads ? 1u : 2u) | (turn < n_quads ? min(255u, L.cnt[L.quad[turn] & 1023u]) : 0u) << 2 & 0xfcu;
        ++tl_turn;
#endif
        if (turn >= n_turns) break;   // wave-uniform: the ballots inside tile_locate and the shuffles of a quad need whole waves
        if (turn < n_quads) {
            const uint32_t e = L.quad[turn];
            const uint32_t s = e & 1023u, qj = (e >> 10) & 63u, prow = e >> 16;
            const uint32_t c = L.cnt[s], ni = (c + lc - 1) / lc;
            const bool one = ni <= 4;                // the run is this one quad: finished here
            const bool fin = one && !emit;
            const uint32_t j = 4 * qj + (uint32_t)tile;
            const uint32_t nh = j < ni ? min(lc, c - lc * j) : 0u;
            const uint32_t run0 = L.off[s] + lc * j;
            const int64_t key = (int64_t)(L.key[s] ^ kBias);
            int64_t slot = -1;
            if constexpr (LOCATED) slot = (int64_t)L.slot[s];
            else {
                bool is_new, full;
                slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, fin && tile == 0, tile, tl, is_new, full);
            }
            const bool upd = fin && slot >= 0 && tile == 0;
            const RowAt at = row_at<GROUPED>(A, gdesc, slot, upd);
            for (uint32_t col = tl; col < dim4; col += 16) {
                const uint64_t o = at.row * dim4 + col;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f), x1 = w, x2 = w;
                if (upd) {   // the run's row is requested before its sources are summed: the update's round trip overlaps with the sum's
                    w = at.values[o]; x1 = at.s1[o];
                    if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
                }
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                sum_sources<src_rec>(L, A, bk, rec_bucket0, run0, nh, dim4, col, sx, sy, sz, sw);
                sx = tiles_sum(sx); sy = tiles_sum(sy); sz = tiles_sum(sz); sw = tiles_sum(sw);
                if (tile != 0) continue;
                if (upd) {
                    opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                } else if (!fin) {   // a quad of a longer run -> one fp64 partial row of this block | a whole run of a split bucket's slab -> its pending record
                    if (one) rec_store_row4(bk.pend_row + ((uint64_t)(rec_out0 + L.run[s]) * dim4 + col) * 4, sx, sy, sz, sw);
                    else {
                        double2* dst = part_lds ? reinterpret_cast<double2*>(&L.prow[prow & (kLdsPartRows - 1)][(col & 15u) * 4])
                                                : reinterpret_cast<double2*>(A.part + ((uint64_t)(L.part_base + prow) * dim4 + col) * 4);
                        dst[0] = make_double2(sx, sy); dst[1] = make_double2(sz, sw);
                    }
                }
            }
            if (one && !fin && lane == 0) {
                rec_store(bk.pend_key + rec_out0 + L.run[s], key);
                if constexpr (LOCATED) rec_store(bk.pend_slot + rec_out0 + L.run[s], slot);
            }
            continue;
        }
        const uint32_t item = (turn - n_quads) * 4 + (uint32_t)tile;
        const bool valid = item < n_items;
        const uint32_t s = valid ? L.items[item] : 0u;
        const uint32_t c = valid ? L.cnt[s] : 0u;   // <= lc: the run is this one item
        const uint32_t run0 = L.off[s];
        const bool fin = !emit && valid;            // finished here: locate the row, update it once
        const bool single = !SPLIT && fin && c == 1;
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
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f), x1 = w, x2 = w;
            if (upd) {
                w = at.values[o]; x1 = at.s1[o];
                if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
            }
            if (single) {   // gradient row (already requested) + the key's row -> update -> store; nothing else is live here
                if (upd) {
                    const float4 g = col == (uint32_t)tl ? make_float4(gpre.x, gpre.y, gpre.z, gpre.w) : A.grads[(uint64_t)src0 * dim4 + col];
                    opt_update4(a, w, x1, x2, g);
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                }
                continue;
            }
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            sum_sources<src_rec>(L, A, bk, rec_bucket0, run0, c, dim4, col, sx, sy, sz, sw);
            if (fin) {
                if (upd) {
                    opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
                    store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                    if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
                }
            } else rec_store_row4(bk.pend_row + ((uint64_t)(rec_out0 + L.run[s]) * dim4 + col) * 4, sx, sy, sz, sw);   // a whole run of a split bucket's slab -> its pending record
        }
        if (!fin && tl == 0) {
            rec_store(bk.pend_key + rec_out0 + L.run[s], key);
            if constexpr (LOCATED) rec_store(bk.pend_slot + rec_out0 + L.run[s], slot);
        }
    }
    if constexpr (!SPLIT) MEE_TL(A, 4);   // this thread's items done
    MEE_TLP(A, L, 2);
    if (n_big == 0) return;   // block-uniform
    // ---- 4. runs of several quads: a wave adds the run's partial rows (written by this block: visible after the barrier), tile u rows u, u + 4, …,
    //         and finishes the run ----
    __syncthreads();
    MEE_TLP(A, L, 3);
    for (uint32_t k = t >> 6; k < n_big; k += kApplyWaves) {   // wave-uniform
        const uint32_t e = L.big[k];
        const uint32_t s = e & 1023u, p0 = (e >> 10) & 4095u, nq = e >> 22;
        const bool fin = !emit;
        const int64_t key = (int64_t)(L.key[s] ^ kBias);
        int64_t slot = -1;
        if constexpr (LOCATED) slot = (int64_t)L.slot[s];
        else {
            bool is_new, full;
            slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, fin && tile == 0, tile, tl, is_new, full);
        }
        const bool upd = fin && slot >= 0 && tile == 0;
        const RowAt at = row_at<GROUPED>(A, gdesc, slot, upd);
        for (uint32_t col = tl; col < dim4; col += 16) {
            const uint64_t o = at.row * dim4 + col;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f), x1 = w, x2 = w;
            if (upd) {
                w = at.values[o]; x1 = at.s1[o];
                if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
            }
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            for (uint32_t q0 = (uint32_t)tile; q0 < nq; q0 += 8) {   // two rows in flight per tile (nq <= 8: one round trip)
                const uint32_t q1 = q0 + 4;
                const double2* r0 = part_lds ? reinterpret_cast<const double2*>(&L.prow[(p0 + q0) & (kLdsPartRows - 1)][(col & 15u) * 4])
                                             : reinterpret_cast<const double2*>(A.part + ((uint64_t)(L.part_base + p0 + q0) * dim4 + col) * 4);
                const double2* r1 = part_lds ? reinterpret_cast<const double2*>(&L.prow[(p0 + min(q1, nq - 1)) & (kLdsPartRows - 1)][(col & 15u) * 4])
                                             : reinterpret_cast<const double2*>(A.part + ((uint64_t)(L.part_base + p0 + min(q1, nq - 1)) * dim4 + col) * 4);
                const double2 a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
                const bool on = q1 < nq;
                sx += a0.x + (on ? a1.x : 0.0); sy += a0.y + (on ? a1.y : 0.0); sz += b0.x + (on ? b1.x : 0.0); sw += b0.y + (on ? b1.y : 0.0);
            }
            sx = tiles_sum(sx); sy = tiles_sum(sy); sz = tiles_sum(sz); sw = tiles_sum(sw);
            if (tile != 0) continue;
            if (upd) {
                opt_update4(a, w, x1, x2, make_float4((float)sx, (float)sy, (float)sz, (float)sw));
                store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
                if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
            } else if (!fin) rec_store_row4(bk.pend_row + ((uint64_t)(rec_out0 + L.run[s]) * dim4 + col) * 4, sx, sy, sz, sw);
        }
        if (!fin && lane == 0) {
            rec_store(bk.pend_key + rec_out0 + L.run[s], key);
            if constexpr (LOCATED) rec_store(bk.pend_slot + rec_out0 + L.run[s], slot);
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
        if (pb + kTiles > A.max_part) { pb = 0u; atomicOr(A.status, (uint32_t)MEE_STATUS_INTERNAL); }   // (see process_slab)
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
                on[u] = j0 + (uint32_t)u * kTiles < R && rec_load(bk.pend_key + beg + j) == key;
                const double* r = bk.pend_row + ((uint64_t)(beg + j) * dim4 + col) * 4;
                lo[u] = make_double2(rec_load(r), rec_load(r + 1)); hi[u] = make_double2(rec_load(r + 2), rec_load(r + 3));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { sx += on[u] ? lo[u].x : 0.0; sy += on[u] ? lo[u].y : 0.0; sz += on[u] ? hi[u].x : 0.0; sw += on[u] ? hi[u].y : 0.0; }
        }
        double2* dst = reinterpret_cast<double2*>(A.part + ((uint64_t)(L.part_base + q) * dim4 + col) * 4);
        dst[0] = make_double2(sx, sy); dst[1] = make_double2(sz, sw);
    }
    __syncthreads();   // the partial rows were written by this block: visible to its tile 0 after the barrier
    int64_t slot;
    if constexpr (LOCATED) slot = rec_load(bk.pend_slot + beg + any_rec);   // every record of the key carries its slot
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

// A batch WITHOUT a split bucket (every batch of a uniform key stream): block = bucket, and the kernel's first lines are all there is to it.
// A SKEWED batch (some bucket holds more than kBucketCap positions, or the partition made buckets for hot keys) is a list of units with a STATIC
// schedule (run_units): blocks [0, n_hash) own a hash bucket, the agents behind them take slab a, then hot bucket a - S; what the agents cannot
// take runs in later rounds over the bucket blocks, last block first.  History: round 3 had 256 spare blocks at the head of the grid that looped over
// the slabs with a stride and re-scanned the bucket totals for every slab (chains ended at 70-79 us on Zipf(1.05)); a claim counter for the next
// unit cost 768 blocks ~15 us of waiting on one word.
//
// The slab machinery needs more registers (111-128) than the bucket path (74-78); the FULL kernel is bounded to 80 (MEE_APPLY_WAVES = 6: three
// 512-thread blocks per CU) and what does not fit spills there only; the LEAN kernel has no scratch at all.  Tried instead: two launches, the second with
// hipExtAnyOrderLaunch so that they share the device — the launches serialise; the slab path as a __noinline__ function — a kernel's register
// allocation covers its callees (400-600 B of stack per lane).
constexpr uint32_t kGroupDescLds = 64;   // members whose descriptors a GROUPED kernel stages in LDS (3 KB): larger groups read them from device memory
// The units of one block.  SKEW = false (a batch without a split bucket): block = bucket, straight-line code, everything the block needs came
// with the kernel's first round trip (runs0, size0).  SKEW = true: the unit list, slabs first; the two are separate instances so that the
// registers the slab machinery needs (111-124; the kernel is bounded to 80: three blocks per CU) never spill in the path of a uniform batch (77).
template <int KIND, int DIM4, bool LOCATED, bool GROUPED, bool SKEW>
__device__ __forceinline__ void run_units(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, const uint32_t parity, SegRuns runs, const uint32_t size0, const uint32_t pre_a, const uint32_t pre_b,
                                          const GroupDesc* gdesc) {
    const uint32_t tot_base = parity * bk.n_buckets_max;
    constexpr bool skew = SKEW;
    uint32_t S = 0, H = 0;   // slabs of split buckets, hot keys' buckets that one block takes whole: the units of a skewed batch beyond its hash buckets
    if constexpr (SKEW) {
        // ONE scan over the bucket totals: the slabs and the positions in front of each thread's buckets (the positions = where a split bucket's
        // pending records begin: records never outnumber positions), left in LDS for whoever resolves a unit later
        const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads;   // buckets [t * per_t, (t + 1) * per_t) belong to thread t
        unsigned long long mine = 0;
        for (uint32_t q = 0; q < per_t; ++q) {
            const uint32_t bb = threadIdx.x * per_t + q;
            // (up to two buckets per thread — batches of up to ~340K keys — came with the kernel's first round trip)
            const uint32_t tt = per_t <= 2 ? (q ? pre_b : pre_a) : bb < A.nbk ? bk.tot[tot_base + bb] : 0u;
            // low word: slabs (20 bits) | hot keys' buckets that one block takes whole (12 bits); high word: positions
            mine += (unsigned long long)(tt > kBucketCap ? (tt + kSlab - 1) / kSlab : 0u) | (unsigned long long)(bb >= A.nbk_hash && tt != 0 && tt <= kBucketCap) << 20 |
                    (unsigned long long)tt << 32;
        }
        unsigned long long total;
        const unsigned long long ex = block_scan_u64<kApplyWaves>(mine, L.wsum, total);
        L.pre_slabs[threadIdx.x] = (uint32_t)ex; L.pre_pos[threadIdx.x] = (uint32_t)(ex >> 32); L.pre_a[threadIdx.x] = pre_a;   // (pre_slabs: slabs | hot keys' whole buckets << 20)
        if (threadIdx.x == 0) { L.pre_slabs[kApplyThreads] = (uint32_t)total; L.pre_pos[kApplyThreads] = (uint32_t)(total >> 32); }
        // (block-uniform.  As vector values the dozen schedule numbers derived from them below are spilled to scratch and reloaded one by one in front
        // of every block's first unit — 3-5 us in the timeline; forced into SGPRs (readfirstlane) the blocks start their units 5 us
        // earlier, the kernel has 28 B less scratch, and the located kernel is SLOWER, 59.1 against 54.7 us, same box: the allocation of the hot
        // loops changes with it.  Measured, kept as it was.)
        S = (uint32_t)total & 0xFFFFFu; H = ((uint32_t)total >> 20) & 0xFFFu;
        // (pinned host word: the units this batch had beyond its hash buckets — the next partition sizes its bucket count by it)
        if (blockIdx.x == 0 && threadIdx.x == 0) report_units(bk, A.h_slabs, S + H);
        MEE_TLS(A, blockIdx.x, 0, wall_clock64());
    }
    // Who takes what.  nbk_hash hash buckets, then one bucket per hot key; G = the grid = one round of the resident block slots.  The blocks
    // [0, nbk_hash) OWN a hash bucket each (its totals and run matrix came with the kernel's first round trip).  The blocks behind them are
    // AGENTS: the partition of a skewed stream makes the hash buckets fewer by the units the latest batch had beyond them, and agent a takes
    // unit a of the list [slabs of the split buckets ..., hot keys' buckets that one block takes whole ...] — the long chains (slab, hand-off,
    // perhaps the bucket's merge) start with the kernel and no hash bucket waits for a slot.  What the agents cannot take — the first skewed
    // batch of a stream has none — : slabs first go to the blocks [0, O) INSTEAD of their buckets, and everything left over (more slabs, hot
    // keys' buckets, the displaced buckets [0, O)) follows in later rounds over the hash buckets' blocks, from the last one downwards.  A static
    // schedule: 768 blocks that claim their next unit from one word within a few microseconds of each other wait ~15 us for it (measured).
    const uint32_t G = gridDim.x, NH = min(A.nbk_hash, G), n_agents = G - NH;
    const uint32_t s_agents = min(S, n_agents), O = min(S - s_agents, NH);        // slabs [0, s_agents): agents; [s_agents, s_agents + O): blocks [0, O)
    const uint32_t h_agents = min(H, n_agents - s_agents);                        // hot keys' whole buckets [0, h_agents): agents
    const uint32_t late_slabs = S - s_agents - O, late_hot = H - h_agents, n_late = late_slabs + late_hot + O + (A.nbk_hash > G ? A.nbk_hash - G : 0u);
    [[maybe_unused]] uint32_t tl_i = 0, tl_units = 0;
    for (uint32_t round = 0;; ++round) {
        // (the thread index is re-read through an empty asm in every turn, as in process_slab: hoisted out of this loop, the per-thread addresses
        // of everything below stayed live across the whole loop and were spilled to scratch — a memory round trip in front of every use)
        uint32_t tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        bool is_slab = false, is_hot = false;
        uint32_t u = blockIdx.x;   // slab number | hot keys' bucket number (among those one block takes whole) | bucket number
        if constexpr (SKEW) {
            if (round == 0) {
                if (u >= NH) {   // an agent
                    const uint32_t a = u - NH;
                    if (a < s_agents) { is_slab = true; u = a; }
                    else if (a - s_agents < h_agents) { is_hot = true; u = a - s_agents; }
                    else continue;   // (block-uniform) nothing in round 0
                } else if (u < O) { is_slab = true; u = s_agents + u; }
            } else {
                if (NH == 0) break;
                const uint32_t j = (round - 1) * NH + (NH - 1 - blockIdx.x);
                if (blockIdx.x >= NH || j >= n_late) break;   // block-uniform
                if (j < late_slabs) { is_slab = true; u = s_agents + O + j; }
                else if (j - late_slabs < late_hot) { is_hot = true; u = h_agents + (j - late_slabs); }
                else if (j - late_slabs - late_hot < O) u = j - late_slabs - late_hot;
                else u = G + (j - late_slabs - late_hot - O);
            }
        }
        ++tl_units;
        if (skew && round == 0) MEE_TLS(A, blockIdx.x, 2, wall_clock64());   // (timeline) the block knows its first unit's kind
        if (SKEW && is_slab) {   // ---- a slab of a split bucket (block-uniform) ----
            __syncthreads();   // (the scan's LDS stores; L.u_* of the unit before)
            if (skew && round == 0) MEE_TLS(A, blockIdx.x, 126, wall_clock64());
            {   // the thread whose buckets hold slab u publishes (bucket, slab of the bucket, size, first pending record)
                const uint32_t lo = L.pre_slabs[tx] & 0xFFFFFu, hi = L.pre_slabs[tx + 1] & 0xFFFFFu;
                if (u >= lo && u < hi) {   // exactly one thread
                    const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads;
                    uint32_t acc = lo, pacc = L.pre_pos[tx];
                    for (uint32_t q = 0; q < per_t; ++q) {
                        const uint32_t bb = tx * per_t + q;
                        // (two buckets per thread at most: the first one's total is in LDS, the second one's is the rest of the thread's positions)
                        const uint32_t tt = per_t <= 2 ? (q ? L.pre_pos[tx + 1] - L.pre_pos[tx] - L.pre_a[tx] : L.pre_a[tx]) : bb < A.nbk ? bk.tot[tot_base + bb] : 0u;
                        const uint32_t x = tt > kBucketCap ? (tt + kSlab - 1) / kSlab : 0u;
                        if (u < acc + x) { L.u_b = bb; L.u_sub = u - acc; L.u_size = tt; L.u_beg = pacc; break; }
                        acc += x; pacc += tt;
                    }
                }
            }
            __syncthreads();
            const uint32_t b = __builtin_amdgcn_readfirstlane(L.u_b), sub = __builtin_amdgcn_readfirstlane(L.u_sub);
            const uint32_t size = __builtin_amdgcn_readfirstlane(L.u_size), beg = __builtin_amdgcn_readfirstlane(L.u_beg);
#if MEE_APPLY_TIMELINE
            if (threadIdx.x == 0) L.tl_w = blockIdx.x * 128 + 4 + 16 * min(tl_i, 5u);
#endif
            MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 0, wall_clock64());
            MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 5, (unsigned long long)b | (unsigned long long)sub << 16 | (unsigned long long)size << 32);
            seg_scan(L, seg_load(A, bk, b, tx), tx);   // the bucket's runs in the partition blocks' slices
            // the slab; the slab that finishes the bucket LAST then runs the merge passes
            process_slab<KIND, DIM4, LOCATED, kEmit, GROUPED>(L, A, bk, sub * kSlab, min(kSlab, size - sub * kSlab), b, beg, parity, gdesc);
            MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 1, wall_clock64());
            // ---- publish this slab's pending records, take a ticket; the slab that draws the last ticket merges the bucket.  The records were
            // written through (rec_store), every wave drains its stores, THEN the ticket; the merger reads them with agent-scope loads
            // (rec_load).  No block ever waits for another.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t tk = __hip_atomic_fetch_add(&bk.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                L.is_last = tk == (size + kSlab - 1) / kSlab - 1;
            }
            __syncthreads();
            MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 2, wall_clock64());
            if (L.is_last) {   // block-uniform
                const uint32_t R = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&bk.pend_cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // every slab added its runs before its ticket
                MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 6, (unsigned long long)R);
                // Merge in passes: records of one key must meet in one pass and a pass holds kSlab records.  Usually the bucket's records fit ONE
                // pass (a hot key's own bucket: one record per slab).  Otherwise the records are taken by the low bits of mix64b(key): `bits0`
                // bits give passes of ~256 records; a pass that still finds more than kSlab splits on one more bit, and a pass whose records all
                // carry ONE key goes to mono_pass (mix64b is a bijection: the splitting ends).
                const bool one_pass = R <= kSlab;
                uint32_t bits0 = 0;
                while (!one_pass && ((uint64_t)kSlab / 2 << bits0) < R && bits0 < 40) ++bits0;
                uint64_t v0 = 0;
                if (threadIdx.x == 0) L.stk_n = 0u;
                for (bool done = false; !done;) {
                    uint32_t m = 0;
                    if (one_pass) {
                        if (tx < R) L.src[tx] = tx;   // (process_slab begins with a barrier)
                        m = R; done = true;
                    }
                    // ---- the next merge pass: pop a hash prefix, collect its records; too many -> split the prefix (or one key: mono_pass) ----
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
                            const int64_t kj = rec_load(bk.pend_key + beg + j);
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
                    if (m == 0) break;
                    MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 3, wall_clock64());   // the pass's records are collected
                    process_slab<KIND, DIM4, LOCATED, kMerge, GROUPED>(L, A, bk, 0, m, b, beg, parity, gdesc);
                    MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 4, wall_clock64());
                }
            }
            MEE_TLS(A, blockIdx.x, 4 + 16 * tl_i + 7, wall_clock64());
            ++tl_i;
        } else {   // ---- a bucket ----
            uint32_t b = u;
            if (SKEW && is_hot) {   // the u-th of the hot keys' buckets that one block takes whole: the thread that owns it in the scan publishes its number
                __syncthreads();
                const uint32_t lo = L.pre_slabs[tx] >> 20, hi = L.pre_slabs[tx + 1] >> 20;
                if (u >= lo && u < hi) {   // exactly one thread
                    const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads;
                    uint32_t acc = lo;
                    for (uint32_t q = 0; q < per_t; ++q) {
                        const uint32_t bb = tx * per_t + q;
                        const uint32_t tt = per_t <= 2 ? (q ? L.pre_pos[tx + 1] - L.pre_pos[tx] - L.pre_a[tx] : L.pre_a[tx]) : bb < A.nbk ? bk.tot[tot_base + bb] : 0u;
                        const uint32_t x = bb >= A.nbk_hash && tt != 0 && tt <= kBucketCap;
                        if (u < acc + x) { L.u_b = bb; break; }
                        acc += x;
                    }
                }
                __syncthreads();
                b = __builtin_amdgcn_readfirstlane(L.u_b);
            }
            uint32_t size = size0;
            if (SKEW && (round != 0 || is_hot)) {   // (else: the block's own bucket, everything came with the kernel's first round trip)
                const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads;
                if (per_t <= 2) {   // the bucket totals are in LDS (the blocks' common scan): an empty bucket — most hot-key buckets — costs no load
                    const uint32_t tb = b / per_t;
                    size = b % per_t ? L.pre_pos[tb + 1] - L.pre_pos[tb] - L.pre_a[tb] : L.pre_a[tb];
                    size = __builtin_amdgcn_readfirstlane(size);
                    if (size != 0 && size <= kBucketCap) runs = seg_load(A, bk, b, tx);
                } else {
                    runs = seg_load(A, bk, b, tx);
                    size = __builtin_amdgcn_readfirstlane(bk.tot[tot_base + b]);
                }
            }
            if (skew) MEE_TLS(A, blockIdx.x, 100 + 4 * min(tl_units - tl_i - 1, 6u), wall_clock64());
            if (size != 0 && size <= kBucketCap) {   // (an empty bucket | a split bucket: its slabs' business)
                MEE_TL(A, 1);   // first round trip done (size known)
#if MEE_APPLY_TIMELINE
                if (threadIdx.x == 0 && !skew) { A.dbg[(uint64_t)blockIdx.x * 8 + 6] = size | (unsigned long long)b << 32; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); A.dbg[(uint64_t)blockIdx.x * 8 + 7] = (unsigned long long)(xcc & 0xf) << 32 | hw; }
#endif
                if (SKEW) seg_scan(L, runs, tx);   // (LEAN: the caller has done it)
                process_slab<KIND, DIM4, LOCATED, kWhole, GROUPED>(L, A, bk, 0, size, b, 0, parity, gdesc);
                MEE_TL(A, 5);
            }
            if (skew) {
                MEE_TLS(A, blockIdx.x, 101 + 4 * min(tl_units - tl_i - 1, 6u), wall_clock64());
                MEE_TLS(A, blockIdx.x, 102 + 4 * min(tl_units - tl_i - 1, 6u), (unsigned long long)b | (unsigned long long)size << 32);
            }
        }
        if (!SKEW || n_late == 0) break;   // (block-uniform: every unit has its block — every batch without a split bucket)
        __syncthreads();   // (the unit's last readers of the LDS tables | the next unit's first writers)
    }
    if (skew) {
        MEE_TLS(A, blockIdx.x, 1, (unsigned long long)(tl_i) | (unsigned long long)(tl_units) << 32);
        MEE_TLS(A, blockIdx.x, 3, wall_clock64());
    }
}

// index within the bucket -> its place in pos / pkey (binary search over the bucket's runs in the partition blocks' slices: seg_scan)
__device__ __forceinline__ uint32_t bucket_entry_at(const ApplyLds& L, uint32_t gi) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t stp = kPartBlocks / 2; stp; stp >>= 1) if (L.seg_first[k + stp] <= gi) k += stp;
    return L.seg_at[k] + (gi - L.seg_first[k]);
}

// how many blocks share a split bucket of the LEAN kernel (its own block and the helpers behind it), and which of them takes a key
constexpr uint32_t kSlowHelpers = 64;
// (Every key of a share costs ~8 us of dependent steps — a scan for the next key, a scan for its rows, the sum, the update —, so the shares are as many as there are
// helpers even for a bucket just beyond a block's reach: a bucket of 1400 positions around a key of 900 holds ~350 other keys — 58 per share with 6 shares (round 4:
// size / 256), 0.5 ms; 6 per share with 65.)
__device__ __forceinline__ uint32_t slow_shares(uint32_t size, uint32_t nbk) { return min(min(kSlowHelpers + 1, nbk), (size + 15u) / 16u); }
__device__ __forceinline__ uint32_t slow_share_of(int64_t key, uint32_t n_sub) { return (uint32_t)__umul64hi(mix64b((uint64_t)key), (uint64_t)n_sub); }

// A split bucket in the LEAN kernel (below): its own block and up to 64 helpers — the blocks of the buckets behind it, once they are done with
// their own — share its KEYS by a second hash (slow_share_of); each takes its keys one at a time in increasing key order — pick the smallest key not yet
// done (one scan of the bucket's keys), add up its gradient rows (a second scan finds them, 1024 entries at a time; 32 tiles sum, shuffles and
// LDS combine), update its row once.  O(distinct keys x bucket size) key reads from L2.  This is the price of the first skewed batches of a stream: the keys
// they report as hot and the slab count left in the pinned host word switch the following batches to the FULL kernel.  No merge tables, no scratch.
// Round 5: the bucket's MAJORITY key — a bucket is split because one key fills it: 14 000 of 15 000 positions — is no longer ONE share's work (15 windows of
// 1024 entries one after the other: 0.3-0.6 ms): every share sums the key's rows in its own windows (window w belongs to share w mod n_sub), leaves ONE fp64
// partial row, draws a ticket, and the share that draws the last one adds the <= 65 partial rows up and updates the key (the pending-record discipline of the
// FULL kernel: write-through stores drained before the ticket, agent-scope loads, no fence).

// the rows of key `cur` (biased) among the bucket's entries, columns [c0, c0 + 16): their fp64 sum is left in L.prow[8] (valid for every thread after the function's last
// barrier); returns how many occurrences were found.  Which entries are looked at: windows win0, win0 + win_step, ... of 1024 entries of the whole bucket (list_n = 0), or
// the list_n <= 1024 entries whose indices L.off lists (the bucket without its majority key: one window)
template <bool LOCATED>
__device__ __forceinline__ uint32_t slow_sum_key(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, const uint32_t size, const unsigned long long cur, const uint32_t c0,
                                                 const uint32_t dim4, const uint32_t win0, const uint32_t win_step, const uint32_t grad_rows, const uint32_t list_n = 0,
                                                 const uint32_t window = kBucketCap /* entries per window, <= kBucketCap */) {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));   // (as in process_slab: nothing derived from the thread index may look invariant across the callers' loops)
    const int lane = t & 63, tile = lane >> 4, tl = lane & 15, wv = t >> 6;
    const uint32_t col = c0 + (uint32_t)tl;
    const bool cok = col < dim4;
    const uint32_t count = list_n ? list_n : size;
    __syncthreads();
    if (t < 16) { double* r = &L.prow[8][tl * 4]; r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = 0.0; }
    uint32_t total = 0;
    for (uint32_t e0 = win0 * window; e0 < count; e0 += win_step * window) {
        __syncthreads();
        if (t == 0) L.n_cand = 0u;
        __syncthreads();
        for (uint32_t i = e0 + t; i < min(count, e0 + window); i += kApplyThreads) {
            const PartEntry en = bk.ent[bucket_entry_at(L, list_n ? L.off[i] : i)];
            if (((unsigned long long)en.key ^ kBias) != cur) continue;
            const uint32_t p = en.pos;
            const uint32_t q = atomicAdd(&L.n_cand, 1u);
            L.src[q] = A.gidx ? min(A.gidx[p], grad_rows - 1) : p;
            if constexpr (LOCATED) if (q == 0) L.slot[0] = A.slots[p];   // every occurrence of a key names the same slot
        }
        __syncthreads();
        const uint32_t nc = L.n_cand;
        total += nc;
        if (nc == 0) continue;   // block-uniform
        double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
        for (uint32_t q0 = (uint32_t)wv * 4 + tile; q0 < nc; q0 += 4 * 4 * kApplyWaves) {   // four rows in flight per tile
            float4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = cok ? A.grads[(uint64_t)L.src[min(q0 + (uint32_t)q * 4 * kApplyWaves, nc - 1)] * dim4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = q0 + (uint32_t)q * 4 * kApplyWaves < nc;
                sx += on ? (double)g[q].x : 0.0; sy += on ? (double)g[q].y : 0.0; sz += on ? (double)g[q].z : 0.0; sw += on ? (double)g[q].w : 0.0;
            }
        }
        sx = tiles_sum(sx); sy = tiles_sum(sy); sz = tiles_sum(sz); sw = tiles_sum(sw);
        if (tile == 0) { double* d = &L.prow[wv][tl * 4]; d[0] = sx; d[1] = sy; d[2] = sz; d[3] = sw; }
        __syncthreads();
        if (t < 16) {
            double* r = &L.prow[8][tl * 4];
            for (int w = 0; w < kApplyWaves; ++w) { const double* d = &L.prow[w][tl * 4]; r[0] += d[0]; r[1] += d[1]; r[2] += d[2]; r[3] += d[3]; }
        }
    }
    __syncthreads();
    return total;
}

// one update of columns [c0, c0 + 16) of `key`'s row from the fp64 sum in L.prow[8] (wave 0; its tile 0 holds the row)
template <int KIND, int DIM4, bool LOCATED, bool GROUPED>
__device__ __forceinline__ void slow_update(ApplyLds& L, const ApplyArgs& A, const OptArgs& a, const GroupDesc* gdesc, const int64_t key, const uint32_t c0, const uint32_t dim4) {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));   // (as in process_slab: nothing derived from the thread index may look invariant across the callers' loops)
    const int lane = t & 63, tile = lane >> 4, tl = lane & 15;
    if (t >= 64) return;
    const uint32_t col = c0 + (uint32_t)tl;
    int64_t slot = -1;
    if constexpr (GROUPED) { const int64_t h = L.slot[0]; slot = h >= 0 && ((uint64_t)h >> kGroupSlotBits) < A.n_tables ? h : -1; }
    else if constexpr (LOCATED) {
        bool stale;
        slot = handle_slot(L.slot[0], A.handle_tag, A.capacity, stale);
        if (stale && lane == 0 && c0 == 0) atomicOr(A.status, (uint32_t)MEE_STATUS_STALE_HANDLE);
    } else {
        bool is_new, full;
        slot = tile_locate<false, false>(const_cast<int64_t*>(A.tkeys), A.nb, key, tile == 0, tile, tl, is_new, full);
    }
    if (tile == 0 && slot >= 0 && col < dim4) {
        const RowAt at = row_at<GROUPED>(A, gdesc, slot, true);
        const uint64_t o = at.row * dim4 + col;
        const double* r = &L.prow[8][tl * 4];
        float4 w = at.values[o], x1 = at.s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND == MEE_OPT_ADAM) x2 = at.s2[o];
        opt_update4(a, w, x1, x2, make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]));
        store_row4(&at.values[o], w); store_row4(&at.s1[o], x1);
        if (KIND == MEE_OPT_ADAM) store_row4(&at.s2[o], x2);
    }
}

// A share of a split bucket (LEAN kernel).  What costs here is WALKING the bucket — 15 000 entries, each found through a binary search over the partition blocks'
// runs.  A bucket is split because ONE key fills it, so: a candidate for its majority key comes from 64 sampled entries (Boyer-Moore votes; every share samples the
// same entries and gets the same answer), and when the candidate holds a quarter of the samples the bucket is worked off BY POSITION:
//   * every share sums the candidate's rows in its own windows of 1024 entries (window w belongs to share w mod n_sub), leaves ONE fp64 partial row and draws a
//     ticket; the share that draws the last one adds the <= 65 partial rows up and updates the key.  A helper reads nothing of the bucket but its windows;
//   * the bucket's own block (share 0) walks the bucket ONCE, four entries in flight per thread: it counts the candidate (the hot-key report) and LISTS the other
//     entries in LDS (L.off: a bucket of 15 000 positions around a key of 14 600 lists 400), and those go through the ordinary bucket path (process_slab<kListed>:
//     hash table, runs, tiles — one pass for all ~350 keys instead of four dependent round trips per key and share).
// Anything else — no dominant candidate, or more than 1024 other entries (two giant keys in one bucket) — is shared by KEY as in round 4: each share takes the keys
// its hash gives it (slow_share_of), one at a time.
constexpr uint32_t kSlowAgree = 16;   // samples (of 64) that must carry the candidate
constexpr uint32_t kSlowWindow = 512;   // entries of a share's window when a bucket is worked off by position (one per thread: one round trip finds the candidate's, <= 16 rows per tile)
constexpr uint32_t kWalk = 4;   // entries in flight per thread in the walk (2 and 4 cost the LEAN kernel the same registers: none beyond its bucket path's)
template <int KIND, int DIM4, bool LOCATED, bool GROUPED, bool OWN /* share 0: the bucket's own block */>
__device__ __forceinline__ void slow_bucket(ApplyLds& L, const ApplyArgs& A, const BucketScratch& bk, const uint32_t b, const uint32_t size, const uint32_t parity, const GroupDesc* gdesc,
                                            const uint32_t sub_arg, const uint32_t n_sub) {
    const uint32_t sub = OWN ? 0u : sub_arg;
    static_assert(!GROUPED || LOCATED, "a group's batch names rows, not keys");
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));   // (as in process_slab: nothing derived from the thread index may look invariant across the callers' loops)
    const uint32_t dim4 = DIM4 ? DIM4 : A.dim4;
    OptArgs a = A.a;
    a.kind = KIND;
    __syncthreads();   // (the caller's wave 0 has just written L.seg_first / L.seg_at: nobody may walk the bucket's entries before that)
    // ---- the candidate: Boyer-Moore votes over 64 entries spread over the bucket (wave 0; merged pairwise with shuffles), then how many samples carry it
    if (t < 64) {
        const PartEntry sm = bk.ent[bucket_entry_at(L, (uint32_t)(((uint64_t)t * size) >> 6))];
        const unsigned long long mine = (unsigned long long)sm.key ^ kBias;
        unsigned long long cand = mine;
        int votes = 1;
#pragma unroll
        for (int d = 32; d; d >>= 1) {
            const unsigned long long oc = (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)cand, d) | (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(cand >> 32), d) << 32;
            const int ov = __shfl_xor(votes, d);
            if (oc == cand) votes += ov; else if (ov > votes) { cand = oc; votes = ov - votes; } else votes -= ov;
        }
        // (lane 0's candidate: always one of the samples)
        cand = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)cand, 0) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(cand >> 32), 0) << 32;
        const unsigned long long agree = __ballot(mine == cand);
        if (t == 0) { L.kmax = cand; L.n_cand = 0u; L.n_items = 0u; L.stk_n = (uint32_t)__popcll(agree); }   // (stk_n: the merge passes' word, free in the LEAN kernel)
        if constexpr (LOCATED) if ((int)t == __ffsll((long long)agree) - 1) L.slot[0] = A.slots[sm.pos];   // (any occurrence: they all name the same slot)
    }
    __syncthreads();
    const unsigned long long cand = L.kmax;
    const int64_t cand_key = (int64_t)(cand ^ kBias);
    const bool by_position = n_sub > 1 && L.stk_n >= kSlowAgree && ((uint64_t)b + 1) * (kSlowHelpers + 1) <= bk.fast_max;   // block-uniform, and the same in every share of the bucket
    // CROWDED: the samples say that what is left beside the candidate will not fit ONE block's list (two giant keys in one bucket: four first skewed batches in ten of
    // a Zipf(1.05) stream have such a bucket) — then the other keys are shared by KEY among all shares (each walks the bucket once and lists ITS keys' entries), the
    // candidate still by position.  (Share 0 alone with an overflowing list: a pass over the whole bucket per key, ~340 keys: 10 ms.)
    const bool crowded = by_position && (uint64_t)(64u - L.stk_n) * size > 64ull * (kBucketCap * 3 / 4);
    const bool by_key = !by_position || crowded;   // the other keys: by key among all shares | all to share 0 (one pass of the bucket path)
    // ---- ONE walk (by position: share 0 alone): count the candidate, list everything else (by key: what this share's hash gives it)
    if (by_key || OWN) {
        uint32_t mine = 0;
        for (uint32_t e0 = t; e0 < size; e0 += kWalk * kApplyThreads) {
            PartEntry en[kWalk];
#pragma unroll
            for (uint32_t q = 0; q < kWalk; ++q) en[q] = bk.ent[bucket_entry_at(L, min(e0 + q * kApplyThreads, size - 1))];
#pragma unroll
            for (uint32_t q = 0; q < kWalk; ++q) {
                const uint32_t e = e0 + q * kApplyThreads;
                if (e >= size) continue;
                if (((unsigned long long)en[q].key ^ kBias) == cand) ++mine;
                else if (!by_key || slow_share_of(en[q].key, n_sub) == sub) { const uint32_t at = atomicAdd(&L.n_items, 1u); if (at < kBucketCap) L.off[at] = e; }
            }
        }
        if (mine) atomicAdd(&L.n_cand, mine);
    }
    __syncthreads();
    const uint32_t n_major = L.n_cand, n_other = L.n_items;
    const uint32_t list_n = n_other <= kBucketCap ? n_other : 0u;   // 0: the others do not fit the list (walk the whole bucket for them)
    if (by_position) {
        // windows of kSlowWindow entries, window w to share w mod n_pos; a bucket with fewer windows than shares leaves the shares behind its windows out altogether
        // (no partial row, no ticket: most helper duties of a batch end here, two round trips in)
        const uint32_t n_pos = min(n_sub, (size + kSlowWindow - 1) / kSlowWindow);
        if (!OWN && sub >= n_pos && !crowded) return;   // (block-uniform)
        if (sub < n_pos) {
        const int64_t slot_handle = LOCATED ? (int64_t)L.slot[0] : 0;
        double* prt = bk.pend_row + ((uint64_t)b * (kSlowHelpers + 1) + sub) * dim4 * 4;
        for (uint32_t c0 = 0; c0 < dim4; c0 += 16) {
            (void)slow_sum_key<LOCATED>(L, A, bk, size, cand, c0, dim4, sub, n_pos, a.grad_rows, 0u, kSlowWindow);
            if (t < 16 && c0 + t < dim4) rec_store_row4(prt + (uint64_t)(c0 + t) * 4, L.prow[8][t * 4], L.prow[8][t * 4 + 1], L.prow[8][t * 4 + 2], L.prow[8][t * 4 + 3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partial row is out before the ticket is drawn
        __syncthreads();
        if (t == 0) L.is_last = __hip_atomic_fetch_add(&bk.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_pos - 1;
        __syncthreads();
        if (L.is_last) {   // block-uniform: the share that finishes last adds the partial rows up and updates the key
            if constexpr (LOCATED) if (t == 0) L.slot[0] = slot_handle;
            const double* all = bk.pend_row + (uint64_t)b * (kSlowHelpers + 1) * dim4 * 4;
            for (uint32_t c0 = 0; c0 < dim4; c0 += 16) {
                __syncthreads();
                if (t < 16 && c0 + t < dim4) {
                    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
                    for (uint32_t j = 0; j < n_pos; ++j) {
                        const double* q = all + ((uint64_t)j * dim4 + c0 + t) * 4;
                        r0 += rec_load(q); r1 += rec_load(q + 1); r2 += rec_load(q + 2); r3 += rec_load(q + 3);
                    }
                    double* r = &L.prow[8][t * 4];
                    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
                }
                __syncthreads();
                slow_update<KIND, DIM4, LOCATED, GROUPED>(L, A, a, gdesc, cand_key, c0, dim4);
            }
        }
        }
        if (OWN && t == 0 && n_major >= A.hot_count) report_hot_key(bk, parity, cand_key);   // its own bucket next time
        if (!crowded) {
            if constexpr (!OWN) return;   // a helper's part ends here
            else {
                if (n_other == 0) return;
                if (list_n) {   // the rest of the bucket is an ordinary bucket's worth of entries: the ordinary path
                    process_slab<KIND, DIM4, LOCATED, kListed, GROUPED>(L, A, bk, 0, list_n, b, 0, parity, gdesc);
                    return;
                }
            }
        }
    } else if (n_major != 0 && slow_share_of(cand_key, n_sub) == sub) {   // the candidate is an ordinary key of this share (it is not in the list)
        for (uint32_t c0 = 0; c0 < dim4; c0 += 16) {
            (void)slow_sum_key<LOCATED>(L, A, bk, size, cand, c0, dim4, 0u, 1u, a.grad_rows);
            slow_update<KIND, DIM4, LOCATED, GROUPED>(L, A, a, gdesc, cand_key, c0, dim4);
        }
        if (t == 0 && n_major >= A.hot_count) report_hot_key(bk, parity, cand_key);
    }
    // ---- the other keys, shared by key: each share takes the keys its hash gives it, one at a time in increasing key order
    // (by position with a list that overflows against the samples' word: share 0 takes them all)
    const uint32_t k_sub = by_key ? sub : 0u, k_n = by_key ? n_sub : 1u;
    const uint32_t walk_n = list_n ? list_n : size;
    bool have_last = false;
    unsigned long long last = 0;
    for (;;) {
        // the next key: the smallest (biased) key beyond the last one
        __syncthreads();
        if (t == 0) L.kmin = ~0ull;
        __syncthreads();
        unsigned long long mn = ~0ull;
        for (uint32_t i = t; i < walk_n; i += kApplyThreads) {
            const int64_t k = bk.ent[bucket_entry_at(L, list_n ? L.off[i] : i)].key;
            const unsigned long long bkey = (unsigned long long)k ^ kBias;
            if ((!have_last || bkey > last) && bkey < mn && bkey != cand && slow_share_of(k, k_n) == k_sub) mn = bkey;
        }
        if (mn != ~0ull) atomicMin(&L.kmin, mn);
        __syncthreads();
        const unsigned long long cur = L.kmin;
        if (cur == ~0ull) break;   // block-uniform
        const int64_t key = (int64_t)(cur ^ kBias);
        // its gradient rows, one group of 64 columns at a time (the LDS rows hold 64: the update is element-wise, column groups are independent), then one update
        uint32_t total = 0;
        for (uint32_t c0 = 0; c0 < dim4; c0 += 16) {
            total = slow_sum_key<LOCATED>(L, A, bk, size, cur, c0, dim4, 0u, 1u, a.grad_rows, list_n);
            slow_update<KIND, DIM4, LOCATED, GROUPED>(L, A, a, gdesc, key, c0, dim4);
        }
        if (t == 0 && total >= A.hot_count) report_hot_key(bk, parity, key);   // its own bucket next time
        last = cur; have_last = true;
    }
}

// Two kernels.  LEAN: block = bucket, the bucket path alone (77 registers, NO scratch) — what every batch of a uniform key stream runs; a
// split bucket, should one turn up, is taken by its own block the slow way (slow_bucket).  FULL: the unit list of the skewed batches (slabs,
// pending records, merges, hot keys' buckets: 111-128 registers bounded to 80, 170-200 B of scratch per lane).  The host picks by what the
// latest batches of the table were (bucket_apply_launch); both are correct for any batch.  One kernel for both was the first form: the
// scratch it declares for the skewed path — never touched by a uniform batch — cost the uniform batches 4-5 us of their 63 (the same bucket
// path: 0 B of scratch 62.5 us, 56 B 63.1, 128-160 B 67-68, 600 B 94: resident waves are limited by the scratch the queue has for them).
// How a kernel gets the table's BucketScratch: LEAN by value (kernel arguments: there when the first instruction runs); FULL through a pointer to the copy
// bucket_scratch_alloc left in device memory — the struct never changes after mee_table_create, and as kernel arguments its ~20 pointers were 40 of the
// 106 SGPRs the FULL kernel has (it spilled SGPRs into VGPR lanes from its first instruction on).
template <bool FULL> struct BkArg { using type = BucketScratch; static __device__ __forceinline__ const BucketScratch& ref(const BucketScratch& a) { return a; } };
template <> struct BkArg<true> { using type = const BucketScratch* __restrict__; static __device__ __forceinline__ const BucketScratch& ref(const BucketScratch* a) { return *a; } };
template <int KIND, int DIM4, bool LOCATED, bool GROUPED, bool FULL>
__global__ __launch_bounds__(kApplyThreads, kApplyWavesPerSimd) void bkt_apply_kernel(ApplyArgs A, typename BkArg<FULL>::type bk_arg) {
    const BucketScratch& bk = BkArg<FULL>::ref(bk_arg);
    __shared__ ApplyLds L;
    const GroupDesc* gdesc = nullptr;
    if constexpr (GROUPED) {   // the members' planes: needed once per work item, so they come out of LDS (the first barrier inside process_slab publishes them)
        __shared__ GroupDesc gd[kGroupDescLds];
        if (A.n_tables <= kGroupDescLds) {
            for (uint32_t j = threadIdx.x; j < A.n_tables; j += kApplyThreads) gd[j] = A.desc[j];
            gdesc = gd;
        }
    }
    // ONE round trip brings everything the block must know before it can fetch its entries: which copy of the totals this batch's partition
    // filled (bk.seq[1], meepo_apply_part.h), whether that partition met a split bucket, this block's bucket total in BOTH copies, and — lanes of
    // wave 0 — the lengths and places of the bucket's runs in the partition blocks' slices.  (As a chain seq -> total -> run lengths these were
    // three dependent loads, 2-3 us of every block's life before its first useful request.)
    MEE_TL(A, 0);
    // (FULL: the grid is one round of the resident block slots; the blocks beyond the hash buckets have no bucket of their own)
    if (FULL && A.part_blocks == 0) return;   // (grid-uniform) no batch: the launch that makes the queue hold this kernel's scratch before a skewed batch needs it (bucket_apply_launch)
    const bool own = blockIdx.x < A.nbk_hash;
    const SegRuns runs = seg_load(A, bk, own ? blockIdx.x : 0u, own ? threadIdx.x : 64u);
    const uint32_t tot0 = own ? bk.tot[blockIdx.x] : 0u, tot1 = own ? bk.tot[bk.n_buckets_max + blockIdx.x] : 0u;
    const uint4 hdr = *reinterpret_cast<const uint4*>(bk.seq);   // seq[0] (unused here), seq[1] = the copy, has_split[0], has_split[1]
    // (FULL: what a skewed batch needs next — the totals of the <= 2 buckets this thread owns in the blocks' common scan, both copies — travels
    // with it: 4 loads of L1-resident lines per thread, instead of a dependent round trip in front of every block of a skewed batch)
    uint32_t pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;
    if (FULL && A.nbk <= 2 * kApplyThreads) {
        const uint32_t per_t = (A.nbk + kApplyThreads - 1) / kApplyThreads, ba = threadIdx.x * per_t, bb = ba + 1;
        if (ba < A.nbk) { pa0 = bk.tot[ba]; pa1 = bk.tot[bk.n_buckets_max + ba]; }
        if (per_t == 2 && bb < A.nbk) { pb0 = bk.tot[bb]; pb1 = bk.tot[bk.n_buckets_max + bb]; }
    }
    const uint32_t parity = __builtin_amdgcn_readfirstlane(hdr.y);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&bk.seq[0], 1u);   // this partition is consumed: the next one fills the other copy
    const uint32_t size0 = __builtin_amdgcn_readfirstlane(parity ? tot1 : tot0);
    if constexpr (FULL) {
        // (block-uniform) the unit list: a batch with a split bucket | a batch partitioned with buckets for hot keys (more buckets than blocks,
        // most of them empty)
        // (a batch without a split bucket and without hot keys' buckets takes the same path: S = H = 0, every block its own bucket.  A second
        // instance of the bucket path for it made the kernel 66-72 KB of code — more than the 64 KB instruction cache two CUs share, with bucket,
        // slab and merge units running side by side on them)
        run_units<KIND, DIM4, LOCATED, GROUPED, true>(L, A, bk, parity, runs, size0, parity ? pa1 : pa0, parity ? pb1 : pb0, gdesc);
    } else {   // LEAN: the batch was partitioned into hash buckets only, block = bucket
        SegRuns own_runs = runs;
        if (__builtin_amdgcn_readfirstlane(parity ? hdr.w : hdr.z) != 0) {   // (block-uniform) the batch has a split bucket somewhere
            // helper duty FIRST: block x is share j >= 1 of the split bucket x - j (mod nbk) — a share is a few windows of the bucket's majority key at most (slow_bucket),
            // and the key's update waits for the last of them: behind the blocks' own buckets (round 5's first form) that was 55 us into the kernel at the earliest.
            // Every wave looks at the same 64 buckets in front of the block's own (one load per lane, L2-resident), so the ballot is the same in every wave and the loop
            // below is block-uniform.
            const uint32_t nbk = A.nbk, j = (threadIdx.x & 63u) + 1u;
            const uint32_t y = (blockIdx.x + nbk - j % nbk) % nbk;
            const uint32_t ty = j < nbk ? bk.tot[parity * bk.n_buckets_max + y] : 0u;
            unsigned long long duty = __ballot(ty > kBucketCap && j < slow_shares(ty, nbk));
            while (duty) {
                const uint32_t jj = (uint32_t)__builtin_ctzll(duty);
                duty &= duty - 1;
                const uint32_t yy = __builtin_amdgcn_readlane(y, jj), tyy = __builtin_amdgcn_readlane(ty, jj);
                __syncthreads();   // (the duty before this one is done with the LDS)
                seg_scan(L, seg_load(A, bk, yy, threadIdx.x), threadIdx.x);
                slow_bucket<KIND, DIM4, LOCATED, GROUPED, false>(L, A, bk, yy, tyy, parity, gdesc, jj + 1u, slow_shares(tyy, nbk));
            }
            __syncthreads();
            own_runs = seg_load(A, bk, blockIdx.x, threadIdx.x);   // (fetched again rather than kept in four registers per thread across the duties)
        }
        // (the run starts go into LDS before the paths part: the runs must not stay live across the split bucket's code, which the compiler lays out in front of the
        // ordinary path)
        seg_scan(L, own_runs, threadIdx.x);
        if (size0 <= kBucketCap) {
            if (blockIdx.x == 0 && threadIdx.x == 0) report_units(bk, A.h_slabs, 0u);   // (a block with a split bucket overwrites it when it is done, much later)
            run_units<KIND, DIM4, LOCATED, GROUPED, false>(L, A, bk, parity, own_runs, size0, 0u, 0u, gdesc);
        } else {
            // the stream is skewed: the FULL kernel from now on, with as many agents as this batch had slabs (the sum over the blocks that met a split bucket; the host
            // reads it when it launches the next batch — the sooner it is there, the better)
            if (threadIdx.x == 0) report_slabs(bk, A.h_slabs, (size0 + kSlab - 1) / kSlab, 0u);
            slow_bucket<KIND, DIM4, LOCATED, GROUPED, true>(L, A, bk, blockIdx.x, size0, parity, gdesc, 0u, slow_shares(size0, A.nbk));
        }
    }
}

// ---- calibration of the bucket pairs' split (see kXcdSplit's successor above) -----------------------------------------------------------
__global__ __launch_bounds__(kApplyThreads, kApplyWavesPerSimd) void xcd_probe_kernel(float4* __restrict__ w, float4* __restrict__ acc, const float4* __restrict__ g,
                                                                                      uint32_t row_mask, uint32_t iters, unsigned long long* __restrict__ out) {
    const uint32_t tile = threadIdx.x >> 4, tl = threadIdx.x & 15;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = wall_clock64();
    uint64_t x = mix64((uint64_t)blockIdx.x * 32 + tile + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        x = mix64(x);
        const uint64_t row = x & row_mask, grow = (x >> 32) & row_mask;
        const f32x4 gq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + grow * 16 + tl);
        float4 a = w[row * 16 + tl], b = acc[row * 16 + tl];
        adagrad1(a.x, b.x, gq.x, 0.01f, 1e-10f); adagrad1(a.y, b.y, gq.y, 0.01f, 1e-10f);
        adagrad1(a.z, b.z, gq.z, 0.01f, 1e-10f); adagrad1(a.w, b.w, gq.w, 0.01f, 1e-10f);
        w[row * 16 + tl] = a; acc[row * 16 + tl] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = wall_clock64() - t0;
        out[2 * blockIdx.x + 1] = xcc & 0xfu;
    }
}

struct XcdCalibration { bool done; uint32_t split; mee_calibration rep; };
static std::mutex g_cal_mu;
static XcdCalibration g_cal[64];

static void calibrate_device(int device, XcdCalibration& c) {
    c.done = true; c.split = 0;
    memset(&c.rep, 0, sizeof c.rep);
    c.rep.struct_size = sizeof c.rep;
    if (const char* env = getenv("MEE_XCD_SPLIT")) {   // (pin it: A/B runs, reproducing a number)
        const int v = atoi(env);
        c.split = v > 0 && v < 1024 ? (uint32_t)v : 0u;
        c.rep.xcd_split = c.split; c.rep.from_env = 1;
        return;
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const uint32_t grid = (uint32_t)cus * kApplyBlocksPerCU, rows = 1u << 18, iters = 40;   // one round of the apply's resident block slots; 64 MB per plane
    float4 *w = nullptr, *acc = nullptr, *g = nullptr;
    unsigned long long* out = nullptr;
    std::vector<unsigned long long> h(2 * grid);
    hipError_t e = hipMalloc((void**)&w, (uint64_t)rows * 256);
    if (e == hipSuccess) e = hipMalloc((void**)&acc, (uint64_t)rows * 256);
    if (e == hipSuccess) e = hipMalloc((void**)&g, (uint64_t)rows * 256);
    if (e == hipSuccess) e = hipMalloc((void**)&out, 2ull * grid * 8);
    if (e == hipSuccess) e = hipMemset(w, 0, (uint64_t)rows * 256);
    if (e == hipSuccess) e = hipMemset(acc, 0x3f, (uint64_t)rows * 256);   // (positive accumulators: no NaN paths)
    if (e == hipSuccess) e = hipMemset(g, 0, (uint64_t)rows * 256);
    double ratio[2] = {1.0, 1.0};
    for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {   // the first launch warms caches and TLBs up and is not looked at
        xcd_probe_kernel<<<grid, kApplyThreads, 0, 0>>>(w, acc, g, rows - 1, iters, out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(h.data(), out, 2ull * grid * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess || rep == 0) continue;
        double sum[2] = {0, 0}, byres[8] = {0}, cnt[2] = {0, 0}, nres[8] = {0};
        uint32_t consistent = 1;
        for (uint32_t b = 0; b < grid; ++b) {
            sum[b & 1] += (double)h[2 * b]; cnt[b & 1] += 1;
            byres[b & 7] += (double)h[2 * b]; nres[b & 7] += 1;
            if (b >= 8 && h[2 * b + 1] != h[2 * (b - 8) + 1]) consistent = 0;   // blocks b and b + 8 share an XCD: the placement the weighting relies on
        }
        ratio[rep - 1] = (sum[1] / cnt[1]) / (sum[0] / cnt[0]);
        c.rep.odd_over_even[rep - 1] = (float)ratio[rep - 1];
        if (rep == 2) for (int r = 0; r < 8; ++r) { c.rep.block_us_by_index_mod_8[r] = (float)(byres[r] / (nres[r] ? nres[r] : 1) * 0.01); c.rep.xcc_of_index_mod_8[r] = (uint32_t)h[2 * r + 1]; }
        c.rep.placement_consistent = c.rep.placement_consistent || rep == 1 ? (c.rep.placement_consistent || rep == 1) && consistent : consistent;
    }
    for (void* p : {(void*)w, (void*)acc, (void*)g, (void*)out}) if (p) (void)hipFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return; }   // no calibration: even halves
    const bool same_side = (ratio[0] > 1.0) == (ratio[1] > 1.0);
    const double lo = fmin(fabs(ratio[0] - 1.0), fabs(ratio[1] - 1.0));
    if (same_side && lo >= 0.05 && c.rep.placement_consistent) {
        const double r = 0.5 * (ratio[0] + ratio[1]);          // odd blocks take r times as long per position: the even bucket gets r / (1 + r) of the pair's range
        double split = 1024.0 * r / (1.0 + r);
        split = split < 448.0 ? 448.0 : split > 576.0 ? 576.0 : split;
        c.split = (uint32_t)(split + 0.5);
    }
    c.rep.xcd_split = c.split;
}

uint32_t xcd_split_for_device(int device) {
    std::lock_guard<std::mutex> lk(g_cal_mu);
    if (device < 0 || device >= 64) return 0u;
    if (!g_cal[device].done) calibrate_device(device, g_cal[device]);
    return g_cal[device].split;
}

}  // namespace mee
extern "C" int mee_device_calibration(int32_t device, mee_calibration* out) {
    if (!out || out->struct_size != sizeof(mee_calibration)) return mee::fail(MEE_ERR_INVALID_ARG, "mee_device_calibration: null argument or struct_size != %zu", sizeof(mee_calibration));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || device >= 64) return mee::fail(MEE_ERR_INVALID_ARG, "mee_device_calibration: no such device %d", device);
    mee::DeviceGuard g(device);
    (void)mee::xcd_split_for_device(device);
    std::lock_guard<std::mutex> lk(mee::g_cal_mu);
    *out = mee::g_cal[device].rep;
    return MEE_OK;
}
namespace mee {

// ---- host side ----------------------------------------------------------------------------------------------------------------------
int bucket_scratch_alloc(mee_table* t) {
    BucketScratch& bk = t->bk;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device);
    bk.slots = (uint32_t)cus * kApplyBlocksPerCU;
    bk.fast_max = t->max_batch;   // every batch the table takes: beyond ~5M keys (kMaxBuckets buckets of ~700) the buckets outgrow kBucketCap and go through their slabs
    // buckets the largest batch can be cut into (+ the hot keys' own): the strides of the totals' two copies and of the run matrices
    bk.n_buckets_max = bucket_count_for_host(bk.fast_max, bk.slots);
    bk.n_buckets_max += kHotCap;
    if (bk.n_buckets_max > kMaxBuckets) bk.n_buckets_max = kMaxBuckets;
    bk.n_buckets_max = (bk.n_buckets_max + 63u) & ~63u;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, uint64_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) t->workspace_bytes += bytes; } };
    alloc((void**)&bk.ent, bk.fast_max * sizeof(PartEntry));
    alloc((void**)&bk.cnt_mat, (uint64_t)kPartBlocksMax * bk.n_buckets_max * 4);
    alloc((void**)&bk.off_mat, (uint64_t)kPartBlocksMax * bk.n_buckets_max * 4);
    alloc((void**)&bk.tot, 2ull * bk.n_buckets_max * 4);
    alloc((void**)&bk.seq, 8 * 4);   // seq[0], seq[1], has_split[0], has_split[1]: ONE line of one page, read by every apply block with ONE load; [4]: the LEAN kernel's slab count; [5]: shadow of the pinned host word; [6]: largest slab count published for this batch (report_slabs)
    bk.has_split = bk.seq ? bk.seq + 2 : nullptr;
    alloc((void**)&bk.hot_key, 2ull * kHotSlots * 8); alloc((void**)&bk.hot_idx, 2ull * kHotSlots * 4); alloc((void**)&bk.hot_n, 2 * 4);
    if (e == hipSuccess) e = hipMemset(bk.hot_key, 0, 2ull * kHotSlots * 8);
    if (e == hipSuccess) e = hipMemset(bk.hot_idx, 0xFF, 2ull * kHotSlots * 4);
    if (e == hipSuccess) e = hipMemset(bk.hot_n, 0, 2 * 4);
    if (e == hipSuccess) e = hipMemset(bk.seq, 0, 8 * 4);
    if (e == hipSuccess) e = hipMemset(bk.tot, 0, 2ull * bk.n_buckets_max * 4);   // (every partition launch zeroes the copy the next one adds to)
    alloc((void**)&bk.pend_cnt, (uint64_t)bk.n_buckets_max * 4);
    alloc((void**)&bk.ticket, (uint64_t)bk.n_buckets_max * 4);
    if (t->optimizer != MEE_OPT_NONE) {   // pending records of split buckets: the sparse-optimizer apply alone (dedup and elections take a bucket of any size whole)
        alloc((void**)&bk.pend_key, bk.fast_max * 8);
        alloc((void**)&bk.pend_slot, bk.fast_max * 8);
        alloc((void**)&bk.pend_row, bk.fast_max * (uint64_t)t->dim * sizeof(double));
    }
    // mee_dedup_sum cuts a hot key's own bucket into windows of 1024 positions that leave one fp64 partial row each (meepo_dedup.hip): at most max_batch / 1024 + kHotCap
    bk.sum_part_rows = (uint32_t)(bk.fast_max / 1024 + kHotCap + 8);
    alloc((void**)&bk.sum_part, (uint64_t)bk.sum_part_rows * t->dim * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&bk.h_slabs, 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e == hipSuccess) { *bk.h_slabs = 0u; e = hipHostGetDevicePointer((void**)&bk.h_slabs_dev, bk.h_slabs, 0); }
    bk.skew_adapt = 1; bk.skew_sticky = 0; bk.kernel_choice = -1;
    bk.xcd_split = t->optimizer != MEE_OPT_NONE && t->value_memory == MEE_MEM_HBM ? xcd_split_for_device(t->device) : 0u;   // (only the apply's blocks care)
    // the device-resident copy the FULL apply kernel reads (everything the DEVICE uses of this struct is fixed from here on; the tuning fields are the host's)
    bk.dev_copy = nullptr;
    alloc((void**)&bk.dev_copy, sizeof(BucketScratch));
    if (e == hipSuccess) e = hipMemcpy(bk.dev_copy, &bk, sizeof(BucketScratch), hipMemcpyHostToDevice);
    // the raw-stream operators' state (mee_table::bk_dd): the same scratch arrays, its own pinned word, hot-key set, parity words and totals
    BucketScratch& dd = t->bk_dd;
    dd = bk;
    dd.dev_copy = nullptr; dd.h_slabs = nullptr; dd.h_slabs_dev = nullptr;
    dd.tot = nullptr; dd.seq = nullptr; dd.hot_key = nullptr; dd.hot_idx = nullptr; dd.hot_n = nullptr;
    alloc((void**)&dd.tot, 2ull * bk.n_buckets_max * 4);
    alloc((void**)&dd.seq, 8 * 4);
    dd.has_split = dd.seq ? dd.seq + 2 : nullptr;
    alloc((void**)&dd.hot_key, 2ull * kHotSlots * 8); alloc((void**)&dd.hot_idx, 2ull * kHotSlots * 4); alloc((void**)&dd.hot_n, 2 * 4);
    if (e == hipSuccess) e = hipMemset(dd.hot_key, 0, 2ull * kHotSlots * 8);
    if (e == hipSuccess) e = hipMemset(dd.hot_idx, 0xFF, 2ull * kHotSlots * 4);
    if (e == hipSuccess) e = hipMemset(dd.hot_n, 0, 2 * 4);
    if (e == hipSuccess) e = hipMemset(dd.seq, 0, 8 * 4);
    if (e == hipSuccess) e = hipMemset(dd.tot, 0, 2ull * bk.n_buckets_max * 4);
    if (e == hipSuccess) e = hipHostMalloc((void**)&dd.h_slabs, 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e == hipSuccess) { *dd.h_slabs = 0u; e = hipHostGetDevicePointer((void**)&dd.h_slabs_dev, dd.h_slabs, 0); }
    if (e != hipSuccess) return fail(MEE_ERR_OUT_OF_MEMORY, "hipMalloc for the apply scratch: %s", hipGetErrorString(e));
    return MEE_OK;
}
void bucket_scratch_free(mee_table* t) {
    BucketScratch& bk = t->bk;
    void* dev[] = {bk.ent, bk.cnt_mat, bk.off_mat, bk.tot, bk.seq, bk.hot_key, bk.hot_idx, bk.hot_n, bk.pend_cnt, bk.ticket, bk.pend_key, bk.pend_slot, bk.pend_row, bk.sum_part};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (bk.h_slabs) (void)hipHostFree(bk.h_slabs);
    if (bk.dev_copy) (void)hipFree(bk.dev_copy);
    BucketScratch& dd = t->bk_dd;
    void* dev_dd[] = {dd.tot, dd.seq, dd.hot_key, dd.hot_idx, dd.hot_n};
    for (void* p : dev_dd) if (p) (void)hipFree(p);
    if (dd.h_slabs) (void)hipHostFree(dd.h_slabs);
}

#if MEE_APPLY_TIMELINE
static unsigned long long* g_dbg = nullptr;
extern "C" int mee_debug_timeline(unsigned long long* host_out, uint64_t n_words) {
    if (!g_dbg) return 1;
    return hipMemcpy(host_out, g_dbg, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif
int bucket_apply_prepare(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st) {
    uint32_t grid, nbk;
    bool full;
    const uint32_t nbk_hash = bucket_count_for(t, n, st, &grid, &nbk, &full);
    t->part_full = full;
    uint32_t blocks, per_block;
    part_geometry(n, kPartThreads, blocks, per_block);
    t->part_blocks = blocks; t->part_per_block = per_block; t->part_nbk = nbk; t->part_nbk_hash = nbk_hash; t->part_grid = grid;
    const bool atom = bucket_totals_by_atomics(blocks, nbk);
    bkt_sort_kernel<<<blocks, kPartThreads, sizeof(PartHot) + nbk * 4, st>>>(d_keys, n, nbk_hash, nbk, per_block, t->bk, &t->ctr->status, t->op, atom, t->bk.xcd_split);
    MEE_HIP(hipGetLastError());
    return atom ? MEE_OK : bucket_totals_launch(t, nbk, blocks, st);
}

// the same partition for another consumer (meepo_dedup.hip: last-wins elections), with the geometry the caller chose
int bucket_apply_prepare_as(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st, uint32_t nbk_hash, uint32_t nbk, uint32_t blocks, uint32_t per_block) {
    const bool atom = bucket_totals_by_atomics(blocks, nbk);
    bkt_sort_kernel<<<blocks, kPartThreads, sizeof(PartHot) + nbk * 4, st>>>(d_keys, n, nbk_hash, nbk, per_block, t->bk, &t->ctr->status, t->op, atom, 0u);
    MEE_HIP(hipGetLastError());
    return atom ? MEE_OK : bucket_totals_launch(t, nbk, blocks, st);
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
    A.part = t->bs.gacc; A.max_part = t->bs.max_part; A.op = t->op; A.a = a; A.h_slabs = t->bk.h_slabs_dev;
#if MEE_APPLY_TIMELINE
    if (!g_dbg) (void)hipMalloc((void**)&g_dbg, (16384 * 8 + 1024 * 128) * 8);
    (void)hipMemsetAsync(g_dbg, 0, (16384 * 8 + 1024 * 128) * 8, st);
    A.dbg = g_dbg;
#endif
    A.desc = d_desc; A.n_tables = n_tables;   // a table group's apply (d_slots = the batch's located rows = its keys; t = the group's scratch table)
    A.hot_count = hot_count_for(n);
    A.nbk = t->part_nbk; A.nbk_hash = t->part_nbk_hash; A.part_blocks = t->part_blocks; A.per_block = t->part_per_block;   // as whoever partitioned this batch left them (a tuning
    // call between the partition and the apply — "apply_bucket_max" — must not change the stride the run matrices were written with)
    // one block per bucket (+ the blocks a skewed stream's slabs need); on a skewed batch the blocks work through a list of units, the slabs of the
    // split buckets first (run_units)
    const bool full = t->part_full;   // as the partition decided: FULL = a skewed stream (hot keys' buckets may exist, the grid is one round of the block slots)
    const uint32_t grid = full ? t->part_grid : A.nbk;   // LEAN: block = bucket
#define BK_FULL (t->bk.dev_copy)
#define BKT(K, D4, LOC) do { if (full) bkt_apply_kernel<K, D4, LOC, false, true><<<grid, kApplyThreads, 0, st>>>(A, BK_FULL); else bkt_apply_kernel<K, D4, LOC, false, false><<<grid, kApplyThreads, 0, st>>>(A, t->bk); } while (0)
#define BKT_L(K, D4) do { if (d_desc) { if (full) bkt_apply_kernel<K, D4, true, true, true><<<grid, kApplyThreads, 0, st>>>(A, BK_FULL); else bkt_apply_kernel<K, D4, true, true, false><<<grid, kApplyThreads, 0, st>>>(A, t->bk); } \
                          else if (d_slots) BKT(K, D4, true); else BKT(K, D4, false); } while (0)
#define BKT_D(K) do { if (t->dim4 == 16) BKT_L(K, 16); else if (t->dim4 == 32) BKT_L(K, 32); else BKT_L(K, 0); } while (0)
    if (a.kind == MEE_OPT_ADAGRAD) BKT_D(MEE_OPT_ADAGRAD); else BKT_D(MEE_OPT_ADAM);
    // The FULL kernel needs scratch memory (124-192 B per lane), the LEAN kernel none.  A queue gets its scratch when the first launch that needs it arrives: the
    // runtime allocates it on the host and re-submits — 147 us measured between a stream's forward and its first FULL apply (the second skewed step of a stream: 230-290 us
    // against 95 from the third on).  So the first LEAN apply a table sees on a stream is followed by ONE empty launch of the FULL kernel (same instance, same grid, no batch:
    // every block returns at its first instruction): the allocation happens behind a kernel that is running anyway, on a batch that is not waiting for it.
    if (!full && t->bk.skew_adapt && t->bk.kernel_choice < 0 && !stream_is_capturing(st)) {
        bool ready = false;
        for (uint32_t i = 0; i < t->full_ready_n; ++i) ready |= t->full_ready[i] == (void*)st;
        if (!ready) {
            for (uint32_t i = 3; i > 0; --i) t->full_ready[i] = t->full_ready[i - 1];
            t->full_ready[0] = (void*)st;
            if (t->full_ready_n < 4) ++t->full_ready_n;
            const bool full = true;
            const uint32_t grid = t->part_grid ? t->part_grid : A.nbk;   // (the FULL kernel's grid: one round of the block slots)
            A.part_blocks = 0;
            if (a.kind == MEE_OPT_ADAGRAD) BKT_D(MEE_OPT_ADAGRAD); else BKT_D(MEE_OPT_ADAM);
        }
    }
#undef BKT_D
#undef BKT_L
#undef BKT
#undef BK_FULL
    MEE_HIP(hipGetLastError());
    return MEE_OK;
}

}  // namespace mee

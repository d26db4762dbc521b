// meepo_table_int.h — what the translation units of the table share: the device-side scratch structures, struct mee_table and the
// optimizer update (SPEC.md §4).  Internal: nothing here is part of the C-ABI (include/meepo_embedding.h).
#pragma once
#include "meepo_device.h"
#include "meepo_host.h"

namespace mee {

struct Counters {            // device-resident, persistent
    uint32_t status;         // sticky MEE_STATUS_* bits
    uint32_t election;       // = the epoch of the latest insert whose batch had a position that found its key present (gates its election kernels)
    uint32_t pad[2];
};
struct OpCounters {          // device-resident, zeroed at the start of each op that uses them
    unsigned long long n_export;  // pairs exported / keys counted
    uint32_t n_part;         // fp64 partial-sum rows reserved by the apply's long runs
    uint32_t pad;
    unsigned long long hist[4];   // mee_probe_histogram: lookups that visit 1 / 2 / 3 / 4 or more buckets
};
struct GroupTable {            // S entries, indexed by h
    // One 16-byte entry per h: ent[2h] = key ^ kBias (0 = empty), ent[2h+1] = the count word below.  Key and count share a 64-byte sector,
    // so the claim and the count store of group_kernel, the count load of the apply's main pass and the release (ONE 16-byte store) touch
    // one line per key instead of two.
    unsigned long long* ent;
    // count word, low half ("lo"):  COUNT: occurrences added by blocks other than the claimer's (atomicAdd) | LAST: 1 + highest position seen by them (atomicMax)
    // count word, high half ("hi"): the same quantity of the block whose CAS claimed the entry — a plain store, no atomic.
    // The pair is one aligned 8-byte word: readers take both halves with ONE load (sv_load), so a reader racing with a release sees
    // either the complete pair or zeros, never a mixture.
    long long* sres;           // lent out as a per-position slot list by insert / remove
    uint64_t smask;
};
__device__ __forceinline__ uint32_t* sv_half(const GroupTable& g, uint32_t h) { return reinterpret_cast<uint32_t*>(g.ent + 2 * (uint64_t)h + 1); }   // [0] = lo, [1] = hi
__device__ __forceinline__ void sv_load(const GroupTable& g, uint32_t h, uint32_t& lo, uint32_t& hi) {
    const unsigned long long w = g.ent[2 * (uint64_t)h + 1];
    lo = (uint32_t)w; hi = (uint32_t)(w >> 32);
}
__device__ __forceinline__ void group_release_entry(const GroupTable& g, uint32_t h) {
    reinterpret_cast<ulonglong2*>(g.ent)[h] = make_ulonglong2(0ull, 0ull);
}
struct BatchScratch {          // max_batch entries, indexed by batch position unless noted
    uint32_t *hidx;            // insert: the group-table index of each position's key; mee_dedup_sum: the sorted source lists of buckets beyond the LDS list
    uint32_t *occ;             // lent to the bucketed machinery as its position list (BucketScratch::pos); the admission pass's skip bytes
    uint8_t* fmask;            // found mask of find_or_insert's first pass when the caller passes none
    double* gacc;              // [max_part][dim] fp64 partial-sum rows of the bucketed apply's long runs (tables with an optimizer)
    uint32_t max_part;         // rows of gacc
};



// ---- the bucketed apply (meepo_apply.hip): per-call scratch, all device resident ----------------------------------------------------
// One entry of a partitioned batch: a key and its batch position, 16 bytes — ONE store when the partition files it, ONE load when a consumer pulls it out of a
// run (rounds 3-4 kept positions and keys in two arrays: a run of 2-5 entries touched a line of each; a bucket's ~128 short runs are where the consumers' first
// phase spends its bytes — 100 MB of line fetches for the 12.6 MB of entries of a 1M-key batch, `profiles/r05_dedup.md`)
struct alignas(16) PartEntry { int64_t key; uint32_t pos; uint32_t pad; };
struct BucketScratch {
    PartEntry* ent;       // [fast_max] the batch's (key, position) entries: partition block k's share of the batch, sorted by hash bucket, in slice k
    uint32_t* cnt_mat;    // [kPartBlocksMax][n_buckets_max] keys of each bucket held by each partition block …
    uint32_t* off_mat;    // … and where that run starts inside the block's slice of pos / pkey
    uint32_t* tot;        // [2][n_buckets_max] keys per bucket (added up by the partition blocks); two copies, used alternately
    uint32_t* has_split;  // [2] some bucket holds more than a block takes whole (kBucketCap): the batch has slab units
    uint32_t* seq;        // [2] [0] partitions consumed so far (its low bit picks the copy the next partition fills), [1] the copy the latest partition filled
    uint32_t* pend_cnt;   // [n_buckets_max] split buckets: pending records appended so far
    uint32_t* ticket;     // [n_buckets_max] split buckets: slabs finished
    int64_t* pend_key;    // [fast_max] pending records of split buckets: key …
    int64_t* pend_slot;   // [fast_max] … the table slot its handle named (located applies: the merge needs no probe) …
    double* pend_row;     // [fast_max][dim] … and the fp64 partial sum of its gradient rows within one slab
    unsigned long long* hot_key;   // [2][kHotSlots] the hot-key set (key ^ kBias, 0 = empty): keys that filled a slab, reported by the apply kernel; two copies like tot
    uint32_t* hot_idx;    // [2][kHotSlots] the number the key was given (its bucket in the next batch: nbk_hash + number); 0xFFFFFFFF: none
    uint32_t* hot_n;      // [2] numbers handed out
    uint32_t *h_slabs, *h_slabs_dev;   // pinned host word (and its device address): slabs of the latest batch's split buckets, written by the apply kernel
    uint32_t skew_adapt;  // tuning ("apply_skew_adapt", default 1): size the next batch's bucket count by h_slabs
    uint32_t skew_sticky; // batches for which the FULL apply kernel stays chosen after the latest skewed batch
    int kernel_choice;    // tuning ("apply_kernel"): -1 = by the stream (default), 0 = always the LEAN kernel, 1 = always the FULL kernel
    uint32_t bucket_max;  // tuning ("apply_bucket_max"): positions per bucket aimed at, at most (0 = the default)
    uint32_t n_buckets_max, slots;   // slots: apply blocks the device keeps resident at once (CUs x blocks per CU): bucket counts are multiples of it
    uint64_t fast_max;    // largest n the bucketed path takes
    uint32_t xcd_split;   // tuning ("apply_xcd_split"): see part_bucket_of
    double* sum_part;     // [sum_part_rows][dim] fp64 partial rows of mee_dedup_sum's hot-key windows (meepo_dedup.hip), one per window unit
    uint32_t sum_part_rows;
    BucketScratch* dev_copy;   // this struct in device memory (what the FULL apply kernel reads instead of 40 SGPRs of kernel arguments)
};

}  // namespace mee

struct mee_table {
    int device;
    uint64_t capacity, nb, max_batch;
    uint32_t dim, dim4, optimizer, initializer, value_memory;
    float default_value, init_acc, init_scale;
    uint64_t init_seed;
    // table planes
    int64_t* keys;
    float *values, *s1, *s2;
    uint32_t* hits;             // per-slot access counter (config.flags & MEE_FLAG_TRACK_HITS), else null
    uint32_t* sketch;           // admission policy (config.flags & MEE_FLAG_ADMISSION): count-min sketch, 3 rows of 2^sketch_log2w counters
    uint32_t sketch_log2w;
    // per-batch scratch: group table (S entries) and per-position arrays (max_batch entries)
    uint64_t S, max_part;       // S: group-table entries; max_part: fp64 partial-sum rows of the apply
    mee::GroupTable g;
    mee::BatchScratch bs;
    mee::Counters* ctr;
    mee::OpCounters* op;
    mee::Counters* h_ctr;       // pinned staging for read-backs
    mee::OpCounters* h_op;
    uint64_t table_bytes, workspace_bytes;
    uint64_t generation;        // bumped whenever the planes move (mee_reserve): cached descriptors (mee_group) re-read them
    uint32_t handle_epoch;      // bumped by every call that can move or free a row (remove / clear / reserve): tags the slot handles of mee_find_located
    // a prepared (grouped + planned) apply waiting for its grads: mee_apply_prepare .. mee_apply_*
    uint64_t prepared_n;
    const int64_t* prepared_keys;
    uint32_t epoch;             // batch number of insert / apply launches (tags insert's election flag; never 0)
    // performance knobs (never change results): see mee_set_tuning()
    int find_rounds;            // keys in flight per tile in the find kernel: 1, 2, 4 or 8
    int find_grid_cap;          // max blocks of the find grid (0 = one pass, no grid-stride loop)
    int find_block;             // threads per block of the find launch (64 / 128; anything else: 256)
    int prepare_debug;          // experiments on the training forward: bit 0 = partition as a launch of its own behind it, bits 8.. = cap on its find blocks
    int find_nt;                // bit0: non-temporal row loads, bit1: non-temporal bucket loads, bit2: plain (cached) out stores;
                                // -1 = auto: cached loads (hot rows of skewed streams stay in L2), cached stores while the
                                // dense output fits the Infinity Cache (<= 128 MB), streaming stores beyond
    // mee_find's own cache policy (find_nt = -1, no per-call hint): the dense outputs of the latest lookups, so that a caller whose result buffers ROTATE — nothing
    // re-reads them from cache — gets streaming stores without having to say so (find_plane; relaxed atomics: concurrent lookups may race on it, it steers
    // nothing but the store instruction's cache hint)
    uint64_t out_ring_ptr[8], out_ring_bytes[8];
    uint32_t out_ring_head;
    mee::BucketScratch bk;      // bucketed apply (null pointers when the table has no optimizer)
    // ... and the same scratch with a skew state of its own (the pinned word, the hot-key set, the partition parity and the totals' two copies) for the operators that see the
    // RAW key stream of a batch — dedup_keys, dedup_sum, assign.  A training step on one table runs them beside an apply over the batch's DISTINCT keys (the sharded paths with
    // pre-exchange aggregation): the apply's "no skew" report sent the next dedup of the (skewed) raw stream back to its first-skewed-batch path, every step.
    mee::BucketScratch bk_dd;
    bool prepared_by_forward;   // the pending partition came with a training forward (mee_find*_located_prepare): a mutator in between drops it
    uint32_t part_nbk_hash;   // ... of which the first part_nbk_hash are hash buckets (the rest: one per hot key)
    bool part_full;           // ... and the apply kernel chosen for it (FULL | LEAN: meepo_apply.hip)
    void* full_ready[4];      // streams whose queue already holds the FULL kernel's scratch (bucket_apply_launch), newest first
    uint32_t full_ready_n;
    uint32_t part_blocks, part_per_block, part_nbk, part_grid;   // bucketed apply: how the latest partition split the batch (blocks, batch positions per block, buckets) and the apply grid that goes with it
};

namespace mee {

// ---- sparse optimizers (SPEC.md §4) ------------------------------------------------------------------------
struct OptArgs {
    uint32_t kind;       // MEE_OPT_*
    float lr, eps;       // adagrad: lr; adam: lr unused (step_size)
    float step_size, omb1, omb2;
    uint32_t grad_rows;  // indexed apply: rows of the grad array (indices are clamped to it: caller data never reads out of bounds)
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

__device__ __forceinline__ void update_row(const OptArgs& a, float4* values, float4* s1, float4* s2, uint64_t o, const float4 g) {
    float4 w = values[o], x1 = s1[o], x2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.kind == MEE_OPT_ADAM) x2 = s2[o];
    opt_update4(a, w, x1, x2, g);
    values[o] = w; s1[o] = x1;
    if (a.kind == MEE_OPT_ADAM) s2[o] = x2;
}
// the bucketed apply path (meepo_apply.hip)
uint32_t xcd_split_for_device(int device);   // the calibrated share of a bucket pair's hash range that goes to the even bucket (0 = even halves)
int bucket_scratch_alloc(mee_table* t);
void bucket_scratch_free(mee_table* t);
int bucket_apply_prepare(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st);
int bucket_apply_discard(mee_table* t, hipStream_t st);
int bucket_apply_launch(mee_table* t, const float* d_grads, uint32_t n, const OptArgs& a, const uint32_t* d_gidx, const int64_t* d_slots, hipStream_t st,
                        const GroupDesc* d_desc = nullptr, uint32_t n_tables = 0);
bool bucket_totals_by_atomics(uint32_t blocks, uint32_t nbk);
int bucket_totals_launch(mee_table* t, uint32_t nbk, uint32_t blocks, hipStream_t st, const mee::BucketScratch* bk = nullptr /* default: the apply's */);
int bucket_apply_prepare_as(mee_table* t, const int64_t* d_keys, uint32_t n, hipStream_t st, uint32_t nbk_hash, uint32_t nbk, uint32_t blocks, uint32_t per_block);
// duplicate elimination and last-wins elections on the same partition (meepo_dedup.hip)
int bucket_dedup_keys(mee_table* t, const int64_t* d_keys, uint32_t n, int64_t* d_uniq, int64_t* d_inverse, int64_t miss_index, hipStream_t st);
int bucket_assign(mee_table* t, float* plane, const int64_t* d_keys, const float* d_values, uint32_t n, uint8_t* d_found, hipStream_t st);
int bucket_dedup_sum(mee_table* t, const int64_t* d_keys, const float* d_grads, uint32_t n, int64_t* d_uniq, float* d_gsum, uint32_t* d_counts, int64_t* d_inverse, int64_t miss_index,
                     hipStream_t st);
uint32_t bucket_count_for(mee_table* t, uint64_t n, hipStream_t st, uint32_t* grid_out = nullptr, uint32_t* nbk_total_out = nullptr, bool* full_out = nullptr,
                          uint32_t slots_of = 0, uint32_t bucket_max_of = 0, mee::BucketScratch* state = nullptr /* whose skew state plans the batch; default: the apply's (t->bk) */);

// ---- host helpers the table's translation units share (meepo_table.hip, meepo_find.hip, meepo_export.hip) ------------------------------
inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
inline int64_t handle_tag_of(const mee_table* t) { return (int64_t)(t->handle_epoch & kHandleEpochMask) << kHandleSlotBits; }
inline const float* plane_of(const mee_table* t, uint32_t plane) { return plane == 0 ? t->values : plane == 1 ? t->s1 : plane == 2 ? t->s2 : nullptr; }
// Allocation of one value/state plane in the table's value memory (HBM, or pinned device-mapped host DRAM for a cold tier)
inline hipError_t plane_alloc(uint32_t value_memory, float** p, uint64_t bytes) {
    return value_memory == MEE_MEM_HOST_PINNED ? hipHostMalloc((void**)p, bytes, hipHostMallocMapped | hipHostMallocPortable)
                                               : hipMalloc((void**)p, bytes);
}
inline void plane_free(uint32_t value_memory, float* p) {
    if (p) { if (value_memory == MEE_MEM_HOST_PINNED) (void)hipHostFree(p); else (void)hipFree(p); }
}
// (meepo_table.hip)
uint64_t next_prime(uint64_t n);                                   // SPEC.md §2: the bucket count is prime
void zero_words(void* p, size_t bytes, hipStream_t st);            // zeroing of the small device-side counter blocks (a kernel, not a memset node)
void fill_keys(int64_t* p, uint64_t n, int64_t v, hipStream_t st); // a key plane set to one value
int check_batch(mee_table* t, size_t n, const char* op, void* stream, bool needs_group_table = true, bool drop_pending = true);
// (meepo_find.hip)
int find_plane(const mee_table* t, const float* plane, float miss_value, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream,
               bool missing_only = false, bool counted = false, bool rows_only = false, int64_t* d_slots_out = nullptr, bool unordered = false,
               bool skip_padding = false, int nt_call = -1);

}  // namespace mee

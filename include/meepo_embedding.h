/*
 * meepo_embedding.h — C-ABI of the MI355X (gfx950) GPU backend for a dynamic lookup-table embedding.
 *
 * Drop-in boundary.  The reference snapshot defines NO interface for this path: its only functional
 * statement is /root/reference/README.md:2 ("A distributed high-performance dynamic lookuptable-style
 * Embedding … Supports GPU, CPU, remote distributed KV (such as Redis), SSD, and other backends").  The verbs
 * below (find / insert / assign / export + sparse Adagrad/Adam apply) are the ones BASELINE.json's north_star
 * names for the GPU backend; each entry point cites README.md:2 as the (only) reference anchor and SPEC.md
 * for the semantics it implements.  A maintainer's binding stub is shown in INTEGRATION.md.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++/torch types.  `stream` is a hipStream_t passed as void*
 *    (NULL = the default stream).
 *  - Every `d_*` pointer is DEVICE memory on the table's device, caller-owned, and must stay valid until the
 *    stream reaches the op.  The library never frees or retains caller buffers.
 *  - All ops are asynchronous and stream-ordered unless marked [syncs].  No op allocates device memory after
 *    mee_table_create() (safe for hipGraph capture except the [syncs] ones).
 *  - Return value: MEE_OK or a negative error code; mee_last_error() gives a thread-local message.
 *    Device-side conditions (table full, reserved key in a batch) set sticky bits read by mee_status().
 *  - Mutators (insert/assign/find_or_insert/apply_*), the duplicate reductions (dedup_keys/dedup_sum: they use the table's per-batch
 *    scratch) AND the [syncs] calls (size/status/export/hits_scan/probe_length: they share the table's counter block and its pinned
 *    read-back word) on ONE table must be ordered
 *    by the caller (same stream or events; from several host threads: one such call at a time per table);
 *    concurrent mee_find* calls on different streams / threads are safe.  n must be ≤ config.max_batch for
 *    every op except mee_find (any n).
 *  - There is no CPU fallback: creating a table without a usable gfx950 device fails with MEE_ERR_NO_DEVICE.
 */
#ifndef MEEPO_EMBEDDING_H
#define MEEPO_EMBEDDING_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEE_ABI_VERSION 2   /* 2: mee_dedup_sum is sync-free and padded (round 5) */

#define MEE_EMPTY_KEY     INT64_MIN       /* SPEC.md §2: reserved, never stored; in a batch it is padding (skipped silently) */
#define MEE_RECLAIMED_KEY (INT64_MIN + 1) /* SPEC.md §2: reserved, the tombstone mee_remove leaves */
#define MEE_BUCKET_WIDTH  16              /* keys per bucket = one 128-byte line */

enum { MEE_OK = 0, MEE_ERR_INVALID_ARG = -1, MEE_ERR_OUT_OF_MEMORY = -2, MEE_ERR_HIP = -3,
       MEE_ERR_NO_DEVICE = -4, MEE_ERR_BATCH_TOO_LARGE = -5, MEE_ERR_UNSUPPORTED = -6,
       MEE_ERR_RCCL = -7 /* an RCCL call failed, or librccl.so.1 could not be loaded */ };

enum { MEE_OPT_NONE = 0, MEE_OPT_ADAGRAD = 1, MEE_OPT_ADAM = 2 };
enum { MEE_INIT_CONSTANT = 0, MEE_INIT_UNIFORM = 1 };
/* MEE_STATUS_STALE_HANDLE: mee_apply_*_located met a slot handle made before the table's latest mee_remove / mee_clear / mee_reserve; such
 * positions receive no update (the slot may hold another key by now).
 * MEE_STATUS_INTERNAL: a per-batch scratch area that is sized so that no batch can exhaust it was exhausted after all (a defect of the library:
 * the batch's updates are not to be trusted); never observed, reported rather than hidden. */
enum { MEE_STATUS_TABLE_FULL = 1u, MEE_STATUS_RESERVED_KEY = 2u, MEE_STATUS_STALE_HANDLE = 4u, MEE_STATUS_INTERNAL = 8u };
/* MEE_MEM_HOST_PINNED: rows in pinned, device-mapped host DRAM, read and written by the same kernels over PCIe — the
 * cold tier of a hot/cold pair (BASELINE configs[4]); see meepoembedding_amd/tiered.py. */
enum { MEE_MEM_HBM = 0, MEE_MEM_HOST_PINNED = 1 };
/* MEE_FLAG_TRACK_HITS: keep a per-slot access counter (4 B/slot) fed by mee_find_counted and read by mee_hits_scan —
 * the statistics a hot/cold placement policy needs. */
/* MEE_FLAG_ADMISSION: keep a count-min sketch (3 x max(2^12, capacity / 16 rounded up to a power of two) 32-bit counters) of how
 * often ABSENT keys were asked for — the state of the admission policy of mee_find_or_insert_admit (SPEC.md §3). */
enum { MEE_FLAG_TRACK_HITS = 1u, MEE_FLAG_ADMISSION = 2u };

typedef struct mee_table  mee_table;  /* one HBM-resident hash table (one shard) */
typedef struct mee_router mee_router; /* workspace for the shard partition / un-permute kernels */
typedef struct mee_p2p    mee_p2p;    /* peer-mapped buffers of the all-to-all-free sharded find */
typedef struct mee_sharded mee_sharded; /* one rank's context of the row-sharded table: local shard + RCCL communicator */
#define MEE_IPC_HANDLE_BYTES 64

typedef struct mee_config {
    uint32_t struct_size;         /* = sizeof(mee_config); ABI guard */
    int32_t  device;              /* HIP device ordinal */
    uint64_t capacity;            /* requested slots; rounded up to 16 x (smallest prime >= capacity/16) (SPEC.md §2) */
    uint32_t dim;                 /* floats per row: multiple of 4, 4..1024 */
    uint32_t optimizer;           /* MEE_OPT_*: which state planes to allocate */
    uint64_t max_batch;           /* largest n of any mutating op: sizes the workspace — with an optimizer ~(56 + 10 dim) B per position of HBM
                                   * (0.7 GB at 1M x dim 64: the pending records of a skewed batch's worst case); see mee_table_info.workspace_bytes */
    float    default_value;       /* fill for rows of absent keys */
    float    initial_accumulator; /* Adagrad acc of a newly inserted key */
    uint32_t initializer;         /* MEE_INIT_*: initial row for find_or_insert */
    float    init_scale;
    uint64_t init_seed;
    uint32_t value_memory;        /* MEE_MEM_*: where the value/state planes live (the key plane is always in HBM) */
    uint32_t flags;               /* MEE_FLAG_* bits, 0 by default */
} mee_config;

typedef struct mee_table_info {
    uint64_t capacity, n_buckets, max_batch;
    uint32_t dim, optimizer;
    uint64_t table_bytes;     /* keys + all planes resident in HBM */
    uint64_t workspace_bytes;
} mee_table_info;

int         mee_abi_version(void);
const char* mee_last_error(void);

/* ---- table lifetime (README.md:2 "GPU … backend"; SPEC.md §2) ------------------------------------------ */
int mee_table_create(const mee_config* cfg, mee_table** out);
int mee_table_destroy(mee_table* t);
int mee_table_info_get(const mee_table* t, mee_table_info* out);
int mee_clear(mee_table* t, void* stream);
/* performance knobs; never change results.  "find_rounds" (keys in flight per 16-lane tile: 1/2/4/8), "find_grid_cap" (max blocks of
 * the find grid, 0 = unbounded), "find_nt" (cache policy of a find: bit 0 streaming row loads, bit 1 streaming bucket loads, bit 2 cached
 * stores of the dense output; -1 = the library's rule: cached loads; cached stores while one call's output is <= 128 MB and the table's latest
 * lookups wrote to one buffer, streaming stores once their output buffers rotate (the distinct buffers of the last eight calls exceed 64 MB);
 * prefer mee_find_ex: the hint with the CALL),
 * "apply_bucket_max" (target positions per bucket of the apply, 1..352; 0 = the library's rule: one bucket per resident block slot, as many
 * rounds as the batch needs; the bucket COUNT never exceeds what the table's scratch was sized for at creation — the default rule at max_batch —, so a
 * small value on a batch near max_batch gives larger buckets than asked for), "apply_kernel" (-1 = the library's choice by the stream's skew, 0 = LEAN,
 * 1 = FULL: meepo_apply.hip), "apply_skew_adapt" (0: the partition ignores the latest batch's skew report), "apply_xcd_split" (1..1023: share per 1024 of a
 * bucket pair's hash range that goes to the even bucket — even and odd XCDs differ in read-modify-write rate —, 0 = even halves; -1 = as calibrated on this
 * device when its first table was created).  Unknown names: MEE_ERR_INVALID_ARG. */
int mee_set_tuning(mee_table* t, const char* name, int value);
/* What the library measured on `device` when the first table with an optimizer was created there (or measures now): the apply's blocks run on the XCD their index
 * picks, and one parity's read-modify-write stream can be slower than the other's.  A probe kernel with the apply's access pattern is timed by block-index parity
 * (three launches, the first discarded); "apply_xcd_split" defaults to the share that equalises the two — 0 (even halves) unless both measured launches put the same
 * parity at least 5 % behind and blocks b and b + 8 were seen on one XCD throughout.  The environment variable MEE_XCD_SPLIT pins the value (0..1023). */
typedef struct mee_calibration {
    uint32_t struct_size;                 /* = sizeof(mee_calibration) */
    uint32_t xcd_split;                   /* the value tables on this device start with (0 = even halves) */
    uint32_t from_env;                    /* 1: MEE_XCD_SPLIT, nothing was measured */
    uint32_t placement_consistent;        /* 1: in both measured launches, blocks b and b + 8 ran on the same XCD */
    float    odd_over_even[2];            /* mean block time of odd-indexed blocks / even-indexed blocks, per measured launch */
    float    block_us_by_index_mod_8[8];  /* last launch: mean block time by block index mod 8 */
    uint32_t xcc_of_index_mod_8[8];       /* last launch: the XCC id blocks 0..7 reported */
} mee_calibration;
int mee_device_calibration(int32_t device, mee_calibration* out);

/* ---- lookup-table operators (README.md:2 "lookuptable-style"; SPEC.md §3) ------------------------------ */
/* out[i,:] = row of keys[i] or default_value; found nullable. */
int mee_find(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);
/* mee_find with the cache policy of THIS call in `flags` (never changes results).  Hints are per request, not per table: two request
 * queues with different access patterns share one table without touching anything the other one's calls read (mee_set_tuning("find_nt")
 * stays as the table's default for callers that pass MEE_FIND_DEFAULT and for mee_find).
 *   MEE_FIND_STREAM_STORES  the dense output is not re-read from cache (result buffers that rotate, outputs beyond the Infinity Cache)
 *   MEE_FIND_CACHED_STORES  the opposite: keep the output cached whatever its size (at most one of the two; neither = the library's rule: cached
 *                           while one call's output is <= 128 MB and the table's latest lookups did not rotate over output buffers)
 *   MEE_FIND_STREAM_ROWS    rows are not looked up again soon (uniform streams over a table far larger than the caches); a skewed stream
 *                           wants its hot rows cached and must not set it
 *   MEE_FIND_STREAM_BUCKETS the same for the 128-byte key lines */
enum { MEE_FIND_DEFAULT = 0u, MEE_FIND_STREAM_STORES = 1u, MEE_FIND_CACHED_STORES = 2u, MEE_FIND_STREAM_ROWS = 4u, MEE_FIND_STREAM_BUCKETS = 8u };
int mee_find_ex(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, uint32_t flags, void* stream);
/* mee_find whose launch is NOT ordered behind earlier work of `stream` (hipExtAnyOrderLaunch): it may begin while previous kernels of
 * the stream are still running, so consecutive lookups on one stream overlap their launch latency (measured on MI355X: 32.3 -> 30.8 us
 * per 256K-key lookup).  The caller guarantees that d_keys is complete before the call is issued to the device and that nothing
 * earlier in the stream still reads or writes d_out / d_found (independent requests with buffers of their own).  Work submitted to the
 * stream AFTER it is ordered behind it as usual.  Same results as mee_find. */
int mee_find_unordered(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);
/* Several lookup requests of ONE table in one launch (a server draining its request queue): request q is exactly
 * mee_find(t, reqs[q].d_keys, reqs[q].n, reqs[q].d_out, reqs[q].d_found) (d_found nullable).  `reqs` is a HOST array of at most 16
 * entries (read during the call; the buffers it names are device memory).  The per-launch latency floor is paid once: four
 * 256K-key requests cost what one 1M-key lookup costs. */
typedef struct mee_find_request { const int64_t* d_keys; size_t n; float* d_out; uint8_t* d_found; } mee_find_request;
int mee_find_many(const mee_table* t, const mee_find_request* reqs, uint32_t count, void* stream);
/* mee_find that also reports where each key lives: d_slots_out[i] = an opaque slot handle, -1 for absent / reserved keys.  The
 * handles feed mee_apply_*_located of the SAME training step (forward find -> backward apply) and stay valid only until the next
 * call that can move or free a row of this table (mee_remove, mee_clear, mee_reserve; inserting OTHER keys is fine).  A handle carries
 * the table's layout epoch (bits 40..61; bits 0..39 = the slot mee_locate reports): mee_apply_*_located skips handles of an earlier
 * epoch and raises MEE_STATUS_STALE_HANDLE, so a kept-too-long handle can never update another key's row. */
int mee_find_located(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream);
/* The training forward: mee_find_located + mee_apply_prepare for the same keys in ONE launch.  The grad-independent half of the step's
 * backward (the partition of the batch by hash bucket: latency-bound, 15-18 us per 256K keys) is run by the launch's first blocks beside the
 * row gather (bound by bytes, twice as long), so the following mee_apply_*_located(d_keys, d_slots_out, …) with the SAME d_keys / n starts
 * with its update kernel.  Same rules as mee_apply_prepare while the prepared apply is pending (mee_apply_discard drops it); tables without
 * an optimizer: plain mee_find_located. */
int mee_find_located_prepare(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream);
/* The same for a growing vocabulary: mee_find_or_insert_located whose first launch also carries the partition (absent keys are then created by
 * its second pass exactly as in mee_find_or_insert_located; d_found = present BEFORE the call, nullable). */
int mee_find_or_insert_located_prepare(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream);
/* second-tier pass after a mee_find on another table (same keys/out/found buffers): positions with d_found[i] == 0
 * that THIS table holds get their row and d_found[i] = 1; every other position is left untouched.  No host sync, no
 * compaction: this is how a hot (HBM) table is backed by a cold (pinned host) one inside one stream. */
int mee_find_missing(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);
/* mee_find (missing_only = 0) or mee_find_missing (missing_only = 1) that also adds 1 to the hit counter of every key
 * it finds.  Meant for SAMPLED calls on a hot table (one atomic per found key) and for every call on a cold table (its
 * hits are PCIe-bound anyway).  d_found is required. */
int mee_find_counted(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int missing_only,
                     void* stream);
/* [syncs] keys whose hit counter lies in [min_hits, max_hits] (at most cap are written; *n_out = how many qualified);
 * reset != 0 zeroes every counter, which starts a new observation window. */
int mee_hits_scan(mee_table* t, uint32_t min_hits, uint32_t max_hits, int reset, int64_t* d_keys_out, size_t cap, size_t* n_out,
                  void* stream);
/* Pooled lookup ("embedding bag"): bag b = d_keys[d_bag_offsets[b] .. d_bag_offsets[b+1]) (n_bags+1 uint64 offsets in DEVICE
 * memory); d_out[b,:] = the rows mee_find would return for the bag's positions, added in position order in fp32
 * (MEE_POOL_SUM) and divided by the bag length (MEE_POOL_MEAN); an empty bag gives zeros.  d_found (nullable) is per KEY,
 * indexed like d_keys; n = number of keys (a host value: it only picks the launch shape — a tile per bag, or a whole wave
 * per bag when the average bag is long).  The sum lives in registers: one output row per bag instead of one per key. */
enum { MEE_POOL_SUM = 0, MEE_POOL_MEAN = 1 };
int mee_find_pooled(const mee_table* t, const int64_t* d_keys, size_t n, const uint64_t* d_bag_offsets, size_t n_bags, float* d_out,
                    uint8_t* d_found, int mode, void* stream);
/* upsert; duplicate keys: last occurrence wins. */
int mee_insert(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, void* stream);
/* overwrite only if present; d_found nullable; duplicates: last occurrence wins. */
int mee_assign(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream);
/* the same two verbs on one plane of the table: 0 = values, 1 = acc | m, 2 = v (state planes read 0 for absent keys).
 * They let a caller move a key together with its optimizer state between tables (hot/cold tiers, re-sharding). */
int mee_find_plane(const mee_table* t, uint32_t plane, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found,
                   void* stream);
int mee_assign_plane(mee_table* t, uint32_t plane, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found,
                     void* stream);
/* delete present keys (their slots become RECLAIMED and are reused by later inserts); d_found nullable. */
int mee_remove(mee_table* t, const int64_t* d_keys, size_t n, uint8_t* d_found, void* stream);
/* find, inserting absent keys with their initial row first; d_found (nullable) = present before the call. */
int mee_find_or_insert(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);
/* mee_find_or_insert that also reports where every key lives now: d_slots_out[i] = the slot of keys[i] after the call (-1 only for
 * reserved keys and for keys that could not be created: table full) — the handles mee_apply_*_located of the same training step
 * takes instead of probing again.  Same validity as mee_find_located's handles: until the table's layout changes. */
int mee_find_or_insert_located(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream);
/* Admission policy: mee_find_or_insert that creates an absent key only once it has been asked for often enough.  Every absent
 * position adds 1 to its key's sketch counters; a key whose estimate — after ALL additions of this batch — is >= min_count is
 * created (initial row, initial optimizer state) and all its occurrences return that row; other absent keys return the default
 * row and stay absent.  d_found (nullable) = present before the call.  Deterministic per batch (SPEC.md §3); a count-min sketch
 * only over-estimates, so a key is never admitted later than after min_count requests.  mee_admission_decay: every counter >>= shift
 * (shift >= 32: zero) — a new observation window. */
int mee_find_or_insert_admit(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, uint32_t min_count, void* stream);
int mee_admission_decay(mee_table* t, uint32_t shift, void* stream);
/* Tiered pairs (a key lives in exactly one of two tables): the same two mutators restricted to the positions whose
 * d_found byte is 0, i.e. keys that an earlier pass found in NEITHER table.  d_found is only read. */
int mee_insert_missing(mee_table* t, const int64_t* d_keys, const float* d_values, size_t n, const uint8_t* d_found, void* stream);
int mee_find_or_insert_missing(mee_table* t, const int64_t* d_keys, size_t n, float* d_out, const uint8_t* d_found, void* stream);
/* [syncs] all stored pairs, unspecified order; d_state1/d_state2 (nullable) receive acc|m and v rows in the
 * same order.  At most `cap` pairs are written; *n_out = number stored in the table. */
int mee_export(const mee_table* t, int64_t* d_keys_out, float* d_values_out, float* d_state1_out,
               float* d_state2_out, size_t cap, size_t* n_out, void* stream);
/* [syncs] the same over the slot range [slot_begin, min(slot_end, capacity)): checkpointing / rehashing a table that
 * fills most of HBM walks it in ranges with bounded scratch (a range of S slots holds at most S pairs); the union over a
 * partition of [0, capacity) is mee_export's set.  *n_out = pairs stored in the range. */
int mee_export_range(const mee_table* t, uint64_t slot_begin, uint64_t slot_end, int64_t* d_keys_out, float* d_values_out,
                     float* d_state1_out, float* d_state2_out, size_t cap, size_t* n_out, void* stream);
/* [syncs] rehash in place to at least new_capacity slots (rounded up like mee_table_create; growing or shrinking): every
 * stored pair, its optimizer planes and hit counter move device-to-device into new planes, then the old ones are freed —
 * old and new planes must fit in memory together.  No observable of SPEC.md §3 changes; capacity below the number of
 * stored keys is MEE_ERR_INVALID_ARG; on any error the table is untouched.  Not concurrent with other calls on `t`. */
int mee_reserve(mee_table* t, uint64_t new_capacity, void* stream);
int mee_size(const mee_table* t, size_t* n_out, void* stream);        /* [syncs] */
int mee_status(const mee_table* t, uint32_t* bits_out, void* stream); /* [syncs] */
int mee_clear_status(mee_table* t, void* stream);
/* probe only: d_slots_out[i] = the slot handle of d_keys[i] (-1 = absent / reserved), d_found nullable.  No row is touched. */
int mee_locate(const mee_table* t, const int64_t* d_keys, size_t n, int64_t* d_slots_out, uint8_t* d_found, void* stream);
/* Base address of one plane (0 = values, 1 = acc | m, 2 = v): row of slot s at ptr + s * row_stride_bytes.  A device pointer for
 * MEE_MEM_HBM tables; for MEE_MEM_HOST_PINNED tables the address is valid on the HOST as well (pinned, device-mapped) — this is what
 * a staged transfer (host-side gather + hipMemcpyAsync on a side stream) of a cold tier reads.  Invalidated by mee_reserve / destroy;
 * the caller orders its accesses against the table's operators. */
int mee_table_plane(const mee_table* t, uint32_t plane, void** ptr_out, uint64_t* row_stride_bytes, uint32_t* value_memory);
/* [syncs] measurement aid (SURVEY.md §8d "mean probe length"): the number of buckets a find visits for d_keys, summed over
 * the batch (reserved keys visit none); divide by n for the mean.  Changes nothing. */
int mee_probe_length(const mee_table* t, const int64_t* d_keys, size_t n, uint64_t* buckets_visited_out, void* stream);
/* [syncs] the same as a histogram (SURVEY.md §5 "metrics"): hist_out[0 .. 3] = how many of the batch's lookups visit 1 / 2 / 3 / 4 or more buckets
 * (reserved keys visit none and are not counted). */
int mee_probe_histogram(const mee_table* t, const int64_t* d_keys, size_t n, uint64_t* hist_out /* [4] */, void* stream);

/* ---- table groups: one launch for the lookups of many tables (a model's embedding collection) ----------------
 * All tables of a group live on one device and have the same dim.  The key batches of the tables are concatenated
 * ("jagged" layout): segment j = d_keys[d_offsets[j] .. d_offsets[j+1]), d_offsets = n_tables+1 non-decreasing uint64 in
 * DEVICE memory, n = total positions (host value; positions outside [d_offsets[0], d_offsets[n_tables]) are left
 * untouched).  Results are identical to mee_find on each table with its segment; the launch latency floor is paid once
 * instead of n_tables times.  Asynchronous on `stream`; after mee_reserve on a member the next call re-reads that
 * table's planes (one stream synchronisation).  The group does not own the tables: destroy it before them.  Calls on one
 * group that use its scratch (mee_group_find_or_insert without d_found, mee_group_apply_*) must be ordered by the caller;
 * mee_find_grouped / mee_group_find_pooled calls may run concurrently on several streams. */
typedef struct mee_group mee_group;
int mee_group_create(mee_table* const* tables, uint32_t n_tables, uint64_t max_apply_batch, mee_group** out);
int mee_group_destroy(mee_group* g);
/* mee_set_tuning for the group's own apply ("apply_bucket_max", "apply_kernel", …: as for a table; groups created with
 * max_apply_batch = 0 have nothing to tune: MEE_ERR_UNSUPPORTED). */
int mee_group_set_tuning(mee_group* g, const char* name, int value);
int mee_find_grouped(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found,
                     void* stream);
/* mee_find_or_insert over the jagged layout (three launches): absent keys are created in their member table with that
 * table's initial row / optimizer state first; d_found (nullable when n <= max_apply_batch) = present before the call. */
int mee_group_find_or_insert(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found,
                             void* stream);
/* One sparse-optimizer step over the same jagged layout (groups created with max_apply_batch >= n; all members have the
 * same optimizer): identical to mee_apply_* on each table with its segment of keys and grads (absent keys ignored,
 * duplicates of a key inside its segment summed in fp64 and applied once), in a fixed number of launches whatever the
 * number of tables.  Status bits (RESERVED_KEY for a tombstone value in a segment) land on the member of that segment. */
int mee_group_apply_adagrad(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr,
                            float eps, void* stream);
int mee_group_apply_adam(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr,
                         float beta1, float beta2, float eps, uint64_t step, void* stream);

/* The embedding-bag collection: every member table serves `bags_per_table` bags (one per sample of the batch), bag b belongs
 * to member b / bags_per_table, d_bag_offsets holds n_tables x bags_per_table + 1 key offsets (device) and d_out one pooled row
 * per bag — mee_find_pooled on every member, in one launch.  Backward: mee_apply_*_indexed on every member in the usual 7
 * launches; d_grad_index[i] = the bag of key position i, d_bag_grads = [n_bags, dim] (pre-scaled by 1/length for MEAN).
 * d_located_out / d_located (nullable, int64[n]): the forward can hand the rows it located (opaque per-position handles) to
 * the backward of the SAME step, which then skips its own probe pass (6 launches).  The handles are only valid while no
 * key of the batch is removed, no member is rehashed and no slot they name is reused: i.e. forward -> backward. */
int mee_group_find_pooled(mee_group* g, const int64_t* d_keys, size_t n, const uint64_t* d_bag_offsets, size_t bags_per_table,
                          float* d_out, uint8_t* d_found, int64_t* d_located_out, int mode, void* stream);
int mee_group_apply_adagrad_pooled(mee_group* g, const int64_t* d_keys, const uint64_t* d_bag_offsets, size_t bags_per_table,
                                   const float* d_bag_grads, const uint32_t* d_grad_index, const int64_t* d_located, size_t n, float lr,
                                   float eps, void* stream);
int mee_group_apply_adam_pooled(mee_group* g, const int64_t* d_keys, const uint64_t* d_bag_offsets, size_t bags_per_table,
                                const float* d_bag_grads, const uint32_t* d_grad_index, const int64_t* d_located, size_t n, float lr,
                                float beta1, float beta2, float eps, uint64_t step, void* stream);

/* ---- sparse optimizers (north_star "sparse-optimizer (Adagrad/Adam) scatter-update"; SPEC.md §4) -------- */
int mee_apply_adagrad(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps,
                      void* stream);
int mee_apply_adam(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1,
                   float beta2, float eps, uint64_t step, void* stream);
/* The same step on the slots mee_find_located reported for these keys in the forward pass of the step: the main pass needs no
 * probe of its own (128 B of bucket line and one dependent memory round trip less per key).  d_keys is still required (duplicate
 * reduction); a handle of -1 (key absent at lookup time) receives no update; out-of-range handles are ignored. */
int mee_apply_adagrad_located(mee_table* t, const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr,
                              float eps, void* stream);
int mee_apply_adam_located(mee_table* t, const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr,
                           float beta1, float beta2, float eps, uint64_t step, void* stream);
/* The same with an indirection on the grads: position i takes row d_grad_index[i] of d_grads ([n_grad_rows, dim]) — the
 * backward of a pooled lookup (every key of a bag receives the bag's grad row; for MEE_POOL_MEAN the caller scales the bag
 * rows by 1/length).  Indices >= n_grad_rows are clamped to the last row: bad caller data never reads out of bounds. */
int mee_apply_adagrad_indexed(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index,
                              size_t n, float lr, float eps, void* stream);
int mee_apply_adam_indexed(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index,
                           size_t n, float lr, float beta1, float beta2, float eps, uint64_t step, void* stream);
/* Optional split of an apply: mee_apply_prepare groups the batch's keys (occurrence counts, per-key position lists) — everything
 * that does not need the grads — so it can run early (e.g. on a side stream beside the forward lookup and the dense
 * model); the following mee_apply_adagrad / mee_apply_adam with the SAME d_keys / n then only streams the updates.
 * While a prepared apply is pending only mee_find*, mee_locate, mee_size/status/export and mee_apply_* are accepted;
 * mee_apply_discard drops it.  The caller orders the two streams (event / wait).  (The partition a training forward leaves —
 * mee_find_located_prepare / mee_find_or_insert_located_prepare — is softer: a mutator or mee_reserve that comes before the backward drops
 * it itself, and the apply that follows partitions its batch again.) */
int mee_apply_prepare(mee_table* t, const int64_t* d_keys, size_t n, void* stream);
int mee_apply_discard(mee_table* t, void* stream);
/* Duplicate-key reduction on its own (SPEC.md §4), sync-free: what a rank runs on its gradients before they travel to the keys' owners, so that
 * one (key, summed row) pair per DISTINCT key crosses xGMI instead of one row per occurrence.  Outputs sized for n, padded like mee_dedup_keys' (below):
 *   d_uniq_out[n]          every distinct non-reserved key exactly once, at an unspecified position; MEE_EMPTY_KEY elsewhere (padding, possibly BETWEEN keys)
 *   d_gsum_out[n, dim]     at a key's position: the rows of its occurrences added up in fp64 and rounded once to fp32 (a key that occurs once: its row, bit for
 *                          bit); rows at padding positions are NOT written.  d_grads / d_gsum_out are nullable together (keys, counts and inverse only);
 *                          dim = the table's dim
 *   d_counts_out[n]        (nullable) occurrences of the key; 0 at padding positions
 *   d_inverse_out[n]       (nullable) index of d_keys[i] in d_uniq_out, or miss_index for reserved keys
 * The number of distinct keys never travels to the host; consumers take the padded arrays at their fixed length n (every operator and
 * mee_partition_padded skip MEE_EMPTY_KEY).  Uses the table's per-batch scratch only (no table row is read): n <= config.max_batch. */
int mee_dedup_sum(mee_table* t, const int64_t* d_keys, const float* d_grads, size_t n, int64_t* d_uniq_out,
                  float* d_gsum_out, uint32_t* d_counts_out, int64_t* d_inverse_out, int64_t miss_index,
                  void* stream);

/* The keys-only, sync-free form (what a sharded lookup of a skewed batch needs before the exchange): every distinct non-reserved
 * key of the batch occurs exactly once in d_uniq_out[0 .. n), at an unspecified position; every other entry is MEE_EMPTY_KEY
 * (padding — it may lie BETWEEN the keys: each hash bucket of the batch fills the front of a slice of its own, so that no block
 * has to reserve its place from a shared counter); d_inverse_out[i] = index of d_keys[i] in d_uniq_out, or miss_index for
 * reserved keys.  The number of distinct keys never travels to the host: consumers take the padded array at its fixed length n
 * (padding is skipped by every operator and by mee_partition_padded). */
int mee_dedup_keys(mee_table* t, const int64_t* d_keys, size_t n, int64_t* d_uniq_out, int64_t* d_inverse_out, int64_t miss_index,
                   void* stream);

/* ---- hashing and shard routing (README.md:2 "distributed"; SPEC.md §1, §5) ----------------------------- */
/* any output nullable: mix64(key), bucket(key, n_buckets), owner(key, n_shards). */
int mee_hash_batch(const int64_t* d_keys, size_t n, uint64_t n_buckets, uint32_t n_shards, uint64_t* d_mix_out,
                   uint64_t* d_bucket_out, uint32_t* d_owner_out, void* stream);
int mee_router_create(int32_t device, uint64_t max_batch, uint32_t n_shards, mee_router** out);
int mee_router_destroy(mee_router* r);
/* stable partition by owner: d_send_keys[n], d_counts[n_shards] (uint64), d_perm[n] (batch position). */
int mee_partition(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts,
                  int64_t* d_perm, void* stream);
/* the same, dropping MEE_EMPTY_KEY positions (padding belongs to no shard): the counts add up to the non-padding keys, only the
 * first sum(counts) entries of d_send_keys / d_perm are written. */
int mee_partition_padded(mee_router* r, const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts,
                         int64_t* d_perm, void* stream);
/* out[perm[q], :] = rows[q, :] (row_bytes multiple of 4); inverse of the partition for returned rows/masks. */
int mee_scatter_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out,
                     void* stream);
/* out[q, :] = rows[perm[q], :] — forward permutation of per-key payloads (values / grads) into send order. */
int mee_gather_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out,
                    void* stream);

/* ---- sharded find over peer-mapped memory (README.md:2 "distributed"; SPEC.md §5) ---------------------------------
 * One context per rank; all ranks use the same n_shards / slots_per_peer / max_batch / dim.  The local buffers
 * (key inbox, destination inbox, fill counts, result rows, found bytes, and the payload-row inbox if any) are exported as HIP IPC handles, the caller
 * exchanges them (any side channel) and connects.  Per lookup, after mee_partition:
 *   mee_p2p_push  stores this rank's keys + batch positions into their owners' inboxes            (xGMI stores)
 *   -- barrier across ranks on the stream --
 *   mee_p2p_find  probes what arrived and stores each row straight into the requester's buffers   (xGMI stores)
 *   -- barrier across ranks on the stream --
 * after which mee_p2p_buffers() rows/found [0, n) hold the result in batch order.  slots_per_peer bounds how many keys
 * one rank may send to one owner per lookup; exceeding it drops keys and sets bit 0 of mee_p2p_status. */
int mee_p2p_create(int32_t device, uint32_t n_shards, uint32_t rank, uint64_t slots_per_peer, uint64_t max_batch, uint32_t dim,
                   int with_payload, mee_p2p** out);
int mee_p2p_destroy(mee_p2p* c);
#define MEE_P2P_BUFFERS 6
int mee_p2p_export(mee_p2p* c, void* handles /* MEE_P2P_BUFFERS x MEE_IPC_HANDLE_BYTES */);
int mee_p2p_connect(mee_p2p* c, const void* all_handles /* n_shards x MEE_P2P_BUFFERS x MEE_IPC_HANDLE_BYTES, rank-major */);
/* result buffers: max_batch + 1 rows / bytes — the spare last one is never written by a lookup; a caller that expands a
 * de-duplicated lookup through an index array parks its "no such key" positions there. */
int mee_p2p_buffers(mee_p2p* c, float** d_out, uint8_t** d_found);
int mee_p2p_push(mee_p2p* c, mee_router* r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts, size_t n,
                 void* stream);
int mee_p2p_find(mee_p2p* c, const mee_table* t, void* stream);
/* Mutators over peer-mapped memory (contexts created with with_payload != 0): push (key, row) pairs into the owners'
 * inboxes; every segment is padded to slots_per_peer with MEE_EMPTY_KEY (= padding, SPEC.md §2), so after the barrier
 * the owner hands its WHOLE inbox — mee_p2p_inbox(): n_shards x slots_per_peer keys and rows, ordered by source rank then
 * batch position — to mee_insert / mee_assign / mee_apply_* with a fixed n; a second barrier frees the inbox.  d_rows may be
 * null: keys only, still padded (the owner runs mee_find_or_insert over the inbox, then mee_p2p_find returns the rows). */
int mee_p2p_push_rows(mee_p2p* c, mee_router* r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts,
                      const float* d_rows, size_t n, void* stream);
int mee_p2p_inbox(mee_p2p* c, int64_t** d_keys, float** d_rows, uint64_t* n_slots);
/* The barrier of the sequences above without a collective library: a one-wave kernel announces this rank's arrival in a
 * flag word on every peer and waits (bounded: 5 s, then bit 1 of mee_p2p_status) until every peer has announced its own.
 * Collective: every rank calls it the same number of times, in the same order relative to its pushes and finds. */
int mee_p2p_barrier(mee_p2p* c, void* stream);
int mee_p2p_status(mee_p2p* c, uint32_t* bits_out, void* stream); /* [syncs]; bit 0: inbox overflow, bit 1: barrier time-out */


/* ---- row-sharded table over RCCL (README.md:2 "distributed"; SPEC.md §5; SURVEY.md §8b "sharded variants take a communicator
 * handle", §8e) -------------------------------------------------------------------------------------------------------------
 * One process per GPU; every rank owns the keys with owner(key, G) == rank in its own mee_table and creates one context over
 * an RCCL communicator of the G ranks.  Every mee_sharded_* operator below is COLLECTIVE: all ranks call it, in the same order,
 * each with its own batch (n may differ per rank, 0 included).  On the caller's stream it runs
 *     mee_partition -> keys (+ value / gradient rows) to their owners: ncclGroupStart, G x ncclSend/ncclRecv, ncclGroupEnd
 *     -> the local table's operator over what arrived, ordered by source rank then batch position (so last-wins holds across
 *     ranks) -> rows and found bytes back in ONE grouped exchange -> un-permute into batch order.
 * Results are those of ONE table holding all shards (SPEC.md §5).  d_* pointers are device memory on the communicator's device.
 *
 * `nccl_comm` is an ncclComm_t passed as void*: the caller's own (borrowed, must outlive the context), or one made with the
 * three helpers below by callers that do not link RCCL themselves (the library binds librccl.so.1 at first use; the environment
 * variable MEE_RCCL_LIB names another library to bind instead — a particular RCCL build, or the test suite's stand-in).
 *
 * Segment layout, fixed at creation:
 *   pad_slack = 0   exact: message sizes come from a counts exchange and ONE host synchronisation per operator [syncs].
 *   pad_slack >= 1  padded: every (source, owner) segment holds max_batch / G * pad_slack + 1024 positions, unused ones padded
 *                   with MEE_EMPTY_KEY; message sizes are constants and nothing returns to the host.  A batch that sends one
 *                   owner more than that loses the surplus keys and sets bit 0 of mee_sharded_status (check it when convenient;
 *                   skewed batches: mee_dedup_keys first, or the exact layout).  Mutators then hand the local table G x segment
 *                   positions per call: create it with max_batch >= that.
 *
 * Errors inside a collective operator: everything a rank needs for an operator is allocated when its context is created (owner-side buffers
 * for G x max_batch arrivals; every rank must pass the same max_batch and pad_slack — checked collectively at creation), so no rank can
 * drop out for memory between two exchange steps.  If an RCCL call itself fails, the context aborts its communicator (ncclCommAbort: peers
 * blocked in the same collective return with an error instead of waiting forever) and every later call on it returns MEE_ERR_RCCL.
 * OWNERSHIP after an abort: a context only borrows its communicator, and ncclCommAbort frees it.  The abort is recorded per communicator:
 * every OTHER context that shares it fails at once with MEE_ERR_RCCL as well (none of them touches the freed communicator again),
 * mee_comm_destroy() of an aborted communicator does nothing and returns MEE_OK, and a caller that passed in an ncclComm_t of its own asks
 * mee_comm_aborted(comm) — 1: aborted and freed, do NOT call ncclCommDestroy on it — before it destroys it. */
#define MEE_COMM_ID_BYTES 128
int mee_comm_unique_id(void* id_out /* MEE_COMM_ID_BYTES */);                    /* ncclGetUniqueId: one rank calls, all ranks share the bytes */
int mee_comm_create(const void* id, uint32_t n_ranks, uint32_t rank, int32_t device, void** comm_out); /* ncclCommInitRank (collective) */
int mee_comm_destroy(void* comm);
int mee_comm_aborted(void* comm);   /* 1: this library aborted (and thereby freed) the communicator after an RCCL error.  The mark belongs to the ADDRESS: it is cleared when a
                                     * context is created on a communicator at that address again (mee_sharded_create*: the caller vouches that a live one is there) and by mee_comm_create */
int mee_sharded_create(mee_table* local, void* nccl_comm, uint64_t max_batch /* largest n of a rank per call: the same on every rank */,
                       double pad_slack, mee_sharded** out);
/* The same with options (BASELINE configs[4]; SURVEY.md §7 lever (a)):
 *   cold   the shard is a hot/cold PAIR: `local` holds its hot keys in HBM, `cold` (a table created with MEE_MEM_HOST_PINNED, same dim and
 *          optimizer, same device) the rest.  A key lives in exactly one of the two.  On the owner side a lookup is mee_find on the hot
 *          table + mee_find_missing on the cold one over the same buffers; assign / remove / apply_* go to both (each ignores keys it does
 *          not hold, found masks are OR-ed); new keys are created in the hot table while it holds fewer than hot_key_limit keys (0 = 3/4 of
 *          its capacity; an upper bound is kept on the host and refreshed with one mee_size — a synchronisation — only when it is hit),
 *          else in the cold one.  Which keys are hot is the caller's policy (mee_find_plane / mee_assign_plane move a key with its state).
 *   MEE_SHARDED_DEDUP   lookups (find, find_or_insert) exchange only the batch's DISTINCT keys: mee_dedup_keys on a scratch table of the
 *          context's own -> the padded unique list is partitioned (padding belongs to no shard) -> keys out, rows back -> every occurrence
 *          takes its key's row.  Optimizer applies (apply_adagrad, apply_adam) send ONE (key, summed gradient row) pair per distinct key of the
 *          rank's batch: mee_dedup_sum adds the rank's rows of a key up in fp64 and rounds once, the owner's apply adds the ranks' partial sums
 *          up in fp64 again — within 1e-6 (relative) of the un-aggregated update, not bit-identical to it (one extra rounding per rank and key).
 *          On skewed key streams the bytes on xGMI then scale with the distinct keys, in both directions of a training step, while the result
 *          counts lookups.  Costs ~0.1-0.2 ms of local work per 1M keys: it pays where the saved link time exceeds that. */
enum { MEE_SHARDED_DEDUP = 1u };
typedef struct mee_sharded_options {
    uint32_t struct_size;     /* = sizeof(mee_sharded_options); ABI guard */
    uint32_t flags;           /* MEE_SHARDED_* bits */
    uint64_t max_batch;       /* largest n of a rank per call: the same on every rank */
    double   pad_slack;       /* 0 = exact segments, >= 1 = padded segments (see above) */
    mee_table* cold;          /* nullable: the cold tier behind `local` */
    uint64_t hot_key_limit;   /* with `cold`: keys the hot table may hold before new keys go cold (0 = 3/4 of its capacity) */
} mee_sharded_options;
int mee_sharded_create_ex(mee_table* local, void* nccl_comm, const mee_sharded_options* options, mee_sharded** out);
int mee_sharded_destroy(mee_sharded* s);
int mee_sharded_info(const mee_sharded* s, uint32_t* n_shards, uint32_t* rank, uint64_t* segment_capacity /* 0 = exact layout */);
int mee_sharded_find(mee_sharded* s, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found /* nullable */, void* stream);
int mee_sharded_find_or_insert(mee_sharded* s, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);
int mee_sharded_insert(mee_sharded* s, const int64_t* d_keys, const float* d_values, size_t n, void* stream);
int mee_sharded_assign(mee_sharded* s, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found, void* stream);
int mee_sharded_remove(mee_sharded* s, const int64_t* d_keys, size_t n, uint8_t* d_found, void* stream);
/* one update per distinct key over everything that reaches a shard.  Exact layout: when more pairs arrive than the local table's max_batch
 * (a skewed step, or G ranks x max_batch against a table made for one rank's batch) the arrivals are split by key range — every key's pairs
 * in one chunk, one apply per chunk (one more host synchronisation; MEE_ERR_BATCH_TOO_LARGE only if a single chunk still exceeds max_batch,
 * i.e. one key dominates).  Padded layout: the local table receives G x segment positions per call, a constant known to every rank: a
 * table with a smaller max_batch is refused on every rank alike, before anything is exchanged. */
int mee_sharded_apply_adagrad(mee_sharded* s, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps, void* stream);
int mee_sharded_apply_adam(mee_sharded* s, const int64_t* d_keys, const float* d_grads, size_t n, float lr, float beta1, float beta2,
                           float eps, uint64_t step, void* stream);
int mee_sharded_size(mee_sharded* s, size_t* n_out, void* stream);       /* [syncs] keys stored over all shards (ncclAllReduce) */
/* [syncs]; bit 0: a padded segment overflowed (the lookups it could not send returned the default row and found = 0; mutators lost the
 * surplus pairs).  Sticky until mee_sharded_clear_status. */
int mee_sharded_status(mee_sharded* s, uint32_t* bits_out, void* stream);
int mee_sharded_clear_status(mee_sharded* s, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEEPO_EMBEDDING_H */

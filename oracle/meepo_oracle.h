/*
 * oracle/meepo_oracle.h — CPU oracle for the dynamic lookup-table embedding.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the checker /
 * the reported CPU baseline.  The product (meepoembedding_amd/csrc → libmeepo_hip.so) never links or calls
 * it and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference: README.md, LICENSE, .gitignore) contains no
 * implementation, tests or golden vectors of this path.  The only upstream anchor is README.md:2
 * ("dynamic lookuptable-style Embedding … Supports GPU, CPU … backends").  This oracle is a plain-C
 * implementation of the in-repo SPEC.md ("in-repo CPU backend"), cross-checked against oracle/pyspec.py
 * (hashing) and torch.optim (optimizer math) — see tests/golden/.
 */
#ifndef MEEPO_ORACLE_H
#define MEEPO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEO_EMPTY_KEY     INT64_MIN
#define MEO_RECLAIMED_KEY (INT64_MIN + 1)
#define MEO_BUCKET_W      16

#define MEO_OPT_NONE    0
#define MEO_OPT_ADAGRAD 1
#define MEO_OPT_ADAM    2

#define MEO_INIT_CONSTANT 0
#define MEO_INIT_UNIFORM  1

#define MEO_STATUS_TABLE_FULL   1u
#define MEO_STATUS_RESERVED_KEY 2u

typedef struct meo_table meo_table;

/* SPEC.md §1 */
uint64_t meo_mix64(uint64_t x);
uint64_t meo_mix64b(uint64_t x);
uint64_t meo_mulhi64(uint64_t a, uint64_t b);
uint64_t meo_bucket(int64_t key, uint64_t n_buckets);
uint64_t meo_step(int64_t key, uint64_t n_buckets);
uint64_t meo_next_prime(uint64_t n);
uint32_t meo_owner(int64_t key, uint32_t n_shards);
void     meo_hash_batch(const int64_t* keys, size_t n, uint64_t n_buckets, uint32_t n_shards,
                        uint64_t* mix_out, uint64_t* bucket_out, uint32_t* owner_out);

/* SPEC.md §2-§3 */
meo_table* meo_create(uint64_t capacity, uint32_t dim, uint32_t optimizer, float default_value,
                      float initial_accumulator, uint32_t initializer, float init_scale, uint64_t init_seed);
void     meo_destroy(meo_table* t);
uint64_t meo_capacity(const meo_table* t);
uint64_t meo_size(const meo_table* t);
uint32_t meo_status(const meo_table* t);
void     meo_clear_status(meo_table* t);
void     meo_clear(meo_table* t);
void     meo_initial_row(const meo_table* t, int64_t key, float* row);

void meo_find(const meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found);
/* same, split over nthreads pthreads (read-only, embarrassingly parallel) — CPU baseline timing */
void meo_find_mt(const meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found, int nthreads);   /* persistent worker pool */
/* bulk load of the synthetic key stream (timed CPU baseline only): distinct keys inserted by nthreads threads; returns keys placed */
uint64_t meo_populate_synth_mt(meo_table* t, uint64_t key_seed, uint64_t start, uint64_t count, uint64_t row_seed, int nthreads);
void meo_insert(meo_table* t, const int64_t* keys, const float* values, size_t n);
void meo_assign(meo_table* t, const int64_t* keys, const float* values, size_t n, uint8_t* found);
void meo_find_plane(const meo_table* t, uint32_t plane, const int64_t* keys, size_t n, float* out, uint8_t* found);
void meo_assign_plane(meo_table* t, uint32_t plane, const int64_t* keys, const float* values, size_t n, uint8_t* found);
void meo_remove(meo_table* t, const int64_t* keys, size_t n, uint8_t* found);
void meo_find_or_insert(meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found);
/* plane: 0 values, 1 acc|m, 2 v.  Returns number written (≤ cap pairs). state_out planes nullable. */
uint64_t meo_export(const meo_table* t, int64_t* keys_out, float* values_out, float* state1_out,
                    float* state2_out, uint64_t cap);

/* SPEC.md §4 */
void meo_apply_adagrad(meo_table* t, const int64_t* keys, const float* grads, size_t n, float lr, float eps);
void meo_apply_adam(meo_table* t, const int64_t* keys, const float* grads, size_t n, float lr, float beta1,
                    float beta2, float eps, uint64_t step);
/* duplicate-key reduction on its own: unique keys in first-occurrence order, g_u as SPEC §4, inverse[i]=u.
 * Returns U. uniq/gsum sized for n. Reserved keys are dropped (inverse = -1). */
uint64_t meo_dedup_sum(const int64_t* keys, const float* grads, size_t n, uint32_t dim, int64_t* uniq,
                       float* gsum, int64_t* inverse, uint32_t* counts);

/* SPEC.md §5: stable partition by owner. */
void meo_partition(const int64_t* keys, size_t n, uint32_t n_shards, int64_t* send_keys, uint64_t* counts,
                   int64_t* perm);

#ifdef __cplusplus
}
#endif
#endif

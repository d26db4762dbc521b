/*
 * oracle/meepo_oracle.c — plain-C CPU oracle ("in-repo CPU backend") for SPEC.md.
 *
 * TEST INFRASTRUCTURE ONLY (see meepo_oracle.h).  PARITY UNPINNED: /root/reference has no implementation of
 * this path (README.md:2 is the only functional statement); every function below cites the SPEC.md section
 * it restates instead of a reference file:line.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() only where SPEC.md writes fma).
 */
#include "meepo_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct meo_table {
    uint64_t capacity, n_buckets, size;
    uint32_t dim, optimizer, initializer, status;
    float default_value, initial_accumulator, init_scale;
    uint64_t init_seed;
    int64_t* keys;
    float *values, *s1, *s2; /* s1 = acc (Adagrad) or m (Adam); s2 = v (Adam) */
};

/* ---- SPEC.md §1 ---------------------------------------------------------------------------------------- */
uint64_t meo_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31; return x;
}
uint64_t meo_mix64b(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33; return x;
}
uint64_t meo_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
uint64_t meo_bucket(int64_t key, uint64_t n_buckets) { return meo_mulhi64(meo_mix64((uint64_t)key), n_buckets); }
uint64_t meo_step(int64_t key, uint64_t n_buckets) { return n_buckets > 1 ? 1 + meo_mulhi64(meo_mix64b((uint64_t)key), n_buckets - 1) : 1; }
uint32_t meo_owner(int64_t key, uint32_t n_shards) { return (uint32_t)meo_mulhi64(meo_mix64b((uint64_t)key), n_shards); }

void meo_hash_batch(const int64_t* keys, size_t n, uint64_t n_buckets, uint32_t n_shards, uint64_t* mix_out,
                    uint64_t* bucket_out, uint32_t* owner_out) {
    for (size_t i = 0; i < n; ++i) {
        if (mix_out) mix_out[i] = meo_mix64((uint64_t)keys[i]);
        if (bucket_out) bucket_out[i] = meo_bucket(keys[i], n_buckets);
        if (owner_out) owner_out[i] = meo_owner(keys[i], n_shards);
    }
}

/* ---- SPEC.md §2 ---------------------------------------------------------------------------------------- */
static int reserved(int64_t k) { return k == MEO_EMPTY_KEY || k == MEO_RECLAIMED_KEY; }

static void fill_f32(float* p, uint64_t n, float v) { for (uint64_t i = 0; i < n; ++i) p[i] = v; }

/* SPEC §2: smallest prime >= n */
uint64_t meo_next_prime(uint64_t n) {
    if (n <= 2) return 2;
    if (!(n & 1)) ++n;
    for (;; n += 2) {
        int prime = 1;
        for (uint64_t d = 3; d * d <= n; d += 2)
            if (n % d == 0) { prime = 0; break; }
        if (prime) return n;
    }
}

meo_table* meo_create(uint64_t capacity, uint32_t dim, uint32_t optimizer, float default_value,
                      float initial_accumulator, uint32_t initializer, float init_scale, uint64_t init_seed) {
    if (capacity == 0 || dim < 4 || dim > 1024 || (dim & 3) || optimizer > MEO_OPT_ADAM) return NULL;
    meo_table* t = (meo_table*)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->n_buckets = meo_next_prime((capacity + MEO_BUCKET_W - 1) / MEO_BUCKET_W);
    t->capacity = t->n_buckets * MEO_BUCKET_W;
    t->dim = dim; t->optimizer = optimizer; t->initializer = initializer;
    t->default_value = default_value; t->initial_accumulator = initial_accumulator;
    t->init_scale = init_scale; t->init_seed = init_seed;
    t->keys = (int64_t*)malloc(t->capacity * sizeof(int64_t));
    t->values = (float*)calloc(t->capacity * dim, sizeof(float));
    if (optimizer != MEO_OPT_NONE) t->s1 = (float*)calloc(t->capacity * dim, sizeof(float));
    if (optimizer == MEO_OPT_ADAM) t->s2 = (float*)calloc(t->capacity * dim, sizeof(float));
    if (!t->keys || !t->values || (optimizer != MEO_OPT_NONE && !t->s1) || (optimizer == MEO_OPT_ADAM && !t->s2)) {
        meo_destroy(t); return NULL;
    }
    for (uint64_t i = 0; i < t->capacity; ++i) t->keys[i] = MEO_EMPTY_KEY;
    return t;
}
void meo_destroy(meo_table* t) {
    if (!t) return;
    free(t->keys); free(t->values); free(t->s1); free(t->s2); free(t);
}
uint64_t meo_capacity(const meo_table* t) { return t->capacity; }
uint64_t meo_size(const meo_table* t) { return t->size; }
uint32_t meo_status(const meo_table* t) { return t->status; }
void meo_clear_status(meo_table* t) { t->status = 0; }
void meo_clear(meo_table* t) {
    for (uint64_t i = 0; i < t->capacity; ++i) t->keys[i] = MEO_EMPTY_KEY;
    t->size = 0;
}

/* SPEC §2 probe sequence. Returns slot or -1. If free_out, *free_out = the slot a new key would take: the first
 * RECLAIMED slot met before the probe ended, else the first EMPTY slot of the bucket that ended it (or -1). */
static int64_t probe(const meo_table* t, int64_t key, int64_t* free_out) {
    uint64_t b = meo_bucket(key, t->n_buckets);
    const uint64_t stride = meo_step(key, t->n_buckets);
    int64_t first_tomb = -1;
    if (free_out) *free_out = -1;
    for (uint64_t step = 0; step < t->n_buckets; ++step) {
        const int64_t* kb = t->keys + b * MEO_BUCKET_W;
        int64_t first_empty = -1;
        for (int j = 0; j < MEO_BUCKET_W; ++j) {
            if (kb[j] == key) return (int64_t)(b * MEO_BUCKET_W + j);
            if (kb[j] == MEO_EMPTY_KEY && first_empty < 0) first_empty = (int64_t)(b * MEO_BUCKET_W + j);
            if (kb[j] == MEO_RECLAIMED_KEY && first_tomb < 0) first_tomb = (int64_t)(b * MEO_BUCKET_W + j);
        }
        if (first_empty >= 0) { if (free_out) *free_out = first_tomb >= 0 ? first_tomb : first_empty; return -1; }
        b += stride; if (b >= t->n_buckets) b -= t->n_buckets;
    }
    if (free_out) *free_out = first_tomb;
    return -1;
}

/* SPEC §3 "Initial row" */
void meo_initial_row(const meo_table* t, int64_t key, float* row) {
    if (t->initializer == MEO_INIT_UNIFORM) {
        for (uint32_t j = 0; j < t->dim; ++j) {
            uint64_t h = meo_mix64((uint64_t)key ^ meo_mix64(t->init_seed + j));
            float u = (float)(h >> 40) * 0x1p-24f;
            float c = 2.0f * u - 1.0f; /* exact */
            row[j] = t->init_scale * c;
        }
    } else {
        fill_f32(row, t->dim, t->default_value);
    }
}

static void init_state(meo_table* t, int64_t slot) {
    if (t->optimizer == MEO_OPT_ADAGRAD) fill_f32(t->s1 + slot * t->dim, t->dim, t->initial_accumulator);
    if (t->optimizer == MEO_OPT_ADAM) {
        fill_f32(t->s1 + slot * t->dim, t->dim, 0.0f);
        fill_f32(t->s2 + slot * t->dim, t->dim, 0.0f);
    }
}

/* place a new key (caller checked absent). Returns slot or -1 on table-full. */
static int64_t place(meo_table* t, int64_t key, int64_t empty_slot) {
    if (empty_slot < 0) { t->status |= MEO_STATUS_TABLE_FULL; return -1; }
    t->keys[empty_slot] = key;
    t->size++;
    init_state(t, empty_slot);
    return empty_slot;
}

/* ---- SPEC.md §3 ---------------------------------------------------------------------------------------- */
static void find_range(const meo_table* t, const int64_t* keys, size_t lo, size_t hi, float* out, uint8_t* found) {
    const uint32_t d = t->dim;
    for (size_t i = lo; i < hi; ++i) {
        int64_t s = reserved(keys[i]) ? -1 : probe(t, keys[i], NULL);
        if (s >= 0) memcpy(out + i * d, t->values + s * d, d * sizeof(float));
        else fill_f32(out + i * d, d, t->default_value);
        if (found) found[i] = s >= 0;
    }
}
void meo_find(const meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found) {
    find_range(t, keys, 0, n, out, found);
}
/* Multi-threaded find (the timed CPU baseline): a PERSISTENT worker pool.  Threads are created once (the pool only grows), wait on a
 * condition variable between batches, and take chunks of kFindChunk keys from a shared counter — so a batch costs one wake-up and no
 * pthread_create/join, and a slow thread (oversubscribed box, NUMA) does not hold the batch up with a fixed 1/N share.  The caller
 * works too.  One batch at a time (the pool has one job slot): calls are serialised by pool.lock. */
enum { kFindChunk = 512 };
struct find_pool {
    pthread_mutex_t lock;      /* one batch at a time */
    pthread_mutex_t m;
    pthread_cond_t wake, done;
    int n_threads;             /* workers created so far */
    int want;                  /* workers that take part in the current batch (ids < want) */
    unsigned long generation;  /* bumped per batch */
    int running;               /* workers still inside the current batch */
    /* the job */
    const meo_table* t; const int64_t* keys; size_t n; float* out; uint8_t* found;
    size_t next;               /* next chunk start (atomic) */
};
static struct find_pool g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
                                  0, 0, 0, 0, NULL, NULL, 0, NULL, NULL, 0};
static void pool_work(struct find_pool* p) {
    for (;;) {
        const size_t lo = __atomic_fetch_add(&p->next, (size_t)kFindChunk, __ATOMIC_RELAXED);
        if (lo >= p->n) break;
        const size_t hi = lo + kFindChunk < p->n ? lo + kFindChunk : p->n;
        find_range(p->t, p->keys, lo, hi, p->out, p->found);
    }
}
static void* pool_thread(void* arg) {
    struct find_pool* p = &g_pool;
    const int id = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&p->m);
    for (;;) {
        while (p->generation == seen) pthread_cond_wait(&p->wake, &p->m);
        seen = p->generation;
        if (id >= p->want) continue;   /* this batch runs with fewer threads */
        pthread_mutex_unlock(&p->m);
        pool_work(p);
        pthread_mutex_lock(&p->m);
        if (--p->running == 0) pthread_cond_signal(&p->done);
    }
    return NULL;
}
void meo_find_mt(const meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found, int nthreads) {
    if (nthreads <= 1 || n <= (size_t)kFindChunk) { find_range(t, keys, 0, n, out, found); return; }
    struct find_pool* p = &g_pool;
    pthread_mutex_lock(&p->lock);
    const int helpers = nthreads - 1;   /* the caller is one of the nthreads */
    pthread_mutex_lock(&p->m);
    while (p->n_threads < helpers) {
        pthread_t th;
        if (pthread_create(&th, NULL, pool_thread, (void*)(intptr_t)p->n_threads) != 0) break;   /* run with what there is */
        pthread_detach(th);
        p->n_threads++;
    }
    p->t = t; p->keys = keys; p->n = n; p->out = out; p->found = found;
    __atomic_store_n(&p->next, (size_t)0, __ATOMIC_RELAXED);
    p->want = helpers < p->n_threads ? helpers : p->n_threads;
    p->running = p->want;
    p->generation++;
    pthread_cond_broadcast(&p->wake);
    pthread_mutex_unlock(&p->m);
    pool_work(p);
    pthread_mutex_lock(&p->m);
    while (p->running > 0) pthread_cond_wait(&p->done, &p->m);
    pthread_mutex_unlock(&p->m);
    pthread_mutex_unlock(&p->lock);
}

/* Bulk load of the synthetic key stream for the timed CPU baseline (SURVEY.md 8d generators, bit-identical to meepoembedding_amd/synth.py):
 *   key_i = mix64(key_seed + (start + i + 1) * GOLDEN),  row[j] = (float)(mix64(key ^ mix64(row_seed + j)) >> 40) * 2^-24 - 0.5.
 * The keys of the stream are distinct, so threads insert disjoint key ranges concurrently and only race for SLOTS: a slot is claimed with a
 * compare-and-swap on the key word (SPEC 2 probe sequence; tables that never saw a remove have no RECLAIMED slots to prefer).  Placement
 * differs from a sequential insert, observables (find, size, sorted export) do not.  Not concurrent with any other call on the table. */
#define MEO_GOLDEN 0x9E3779B97F4A7C15ull
struct pop_job { meo_table* t; uint64_t key_seed, row_seed, lo, hi; uint64_t placed; };
static void* pop_thread(void* arg) {
    struct pop_job* j = (struct pop_job*)arg;
    meo_table* t = j->t;
    const uint32_t d = t->dim;
    uint64_t cj[1024];
    for (uint32_t c = 0; c < d; ++c) cj[c] = meo_mix64(j->row_seed + c);
    for (uint64_t i = j->lo; i < j->hi; ++i) {
        const int64_t key = (int64_t)meo_mix64(j->key_seed + (i + 1) * MEO_GOLDEN);
        if (reserved(key)) continue;
        uint64_t b = meo_bucket(key, t->n_buckets);
        const uint64_t stride = meo_step(key, t->n_buckets);
        int64_t slot = -1;
        for (uint64_t step = 0; step < t->n_buckets && slot < 0; ++step) {
            int64_t* kb = t->keys + b * MEO_BUCKET_W;
            for (int q = 0; q < MEO_BUCKET_W && slot < 0; ++q) {
                int64_t cur = __atomic_load_n(&kb[q], __ATOMIC_RELAXED);
                if (cur == key) slot = (int64_t)(b * MEO_BUCKET_W + q);   /* cannot happen for a distinct stream; kept for safety */
                else if (cur == MEO_EMPTY_KEY) {
                    int64_t expect = MEO_EMPTY_KEY;
                    if (__atomic_compare_exchange_n(&kb[q], &expect, key, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) { slot = (int64_t)(b * MEO_BUCKET_W + q); j->placed++; }
                    else if (expect == key) slot = (int64_t)(b * MEO_BUCKET_W + q);
                    else --q;   /* another key took it: look at the same slot again (it is occupied now) */
                }
            }
            b += stride; if (b >= t->n_buckets) b -= t->n_buckets;
        }
        if (slot < 0) continue;   /* table full: reported through the return value */
        float* row = t->values + (uint64_t)slot * d;
        for (uint32_t c = 0; c < d; ++c) row[c] = (float)(meo_mix64((uint64_t)key ^ cj[c]) >> 40) * 0x1p-24f - 0.5f;
        init_state(t, slot);
    }
    return NULL;
}
uint64_t meo_populate_synth_mt(meo_table* t, uint64_t key_seed, uint64_t start, uint64_t count, uint64_t row_seed, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    struct pop_job* jobs = (struct pop_job*)malloc(sizeof(struct pop_job) * (size_t)nthreads);
    int made = 0;
    for (int k = 0; k < nthreads; ++k) {
        jobs[k] = (struct pop_job){t, key_seed, row_seed, start + count * (uint64_t)k / (uint64_t)nthreads, start + count * (uint64_t)(k + 1) / (uint64_t)nthreads, 0};
        if (k + 1 < nthreads && pthread_create(&th[made], NULL, pop_thread, &jobs[k]) == 0) ++made;
        else { pop_thread(&jobs[k]); }   /* the last share (or one whose thread could not be made) runs here */
    }
    /* threads were made for the first `made` jobs that succeeded in order; jobs run inline have finished already */
    for (int k = 0; k < made; ++k) pthread_join(th[k], NULL);
    uint64_t placed = 0;
    for (int k = 0; k < nthreads; ++k) placed += jobs[k].placed;
    t->size += placed;
    free(th); free(jobs);
    return placed;
}

static int has_reserved(meo_table* t, int64_t k) {
    if (k == MEO_RECLAIMED_KEY) t->status |= MEO_STATUS_RESERVED_KEY; /* EMPTY is padding: skipped silently (SPEC §2) */
    return reserved(k);
}

/* sequential batch order makes "last occurrence wins" fall out naturally */
void meo_insert(meo_table* t, const int64_t* keys, const float* values, size_t n) {
    const uint32_t d = t->dim;
    for (size_t i = 0; i < n; ++i) {
        if (has_reserved(t, keys[i])) continue;
        int64_t e, s = probe(t, keys[i], &e);
        if (s < 0) s = place(t, keys[i], e);
        if (s >= 0) memcpy(t->values + s * d, values + i * d, d * sizeof(float));
    }
}
void meo_assign(meo_table* t, const int64_t* keys, const float* values, size_t n, uint8_t* found) {
    const uint32_t d = t->dim;
    for (size_t i = 0; i < n; ++i) {
        int64_t s = has_reserved(t, keys[i]) ? -1 : probe(t, keys[i], NULL);
        if (s >= 0) memcpy(t->values + s * d, values + i * d, d * sizeof(float));
        if (found) found[i] = s >= 0;
    }
}
static float* plane_ptr(const meo_table* t, uint32_t plane) { return plane == 0 ? t->values : plane == 1 ? t->s1 : plane == 2 ? t->s2 : NULL; }
void meo_find_plane(const meo_table* t, uint32_t plane, const int64_t* keys, size_t n, float* out, uint8_t* found) {
    const uint32_t d = t->dim;
    const float* p = plane_ptr(t, plane);
    for (size_t i = 0; i < n; ++i) {
        int64_t s = (!p || reserved(keys[i])) ? -1 : probe(t, keys[i], NULL);
        if (s >= 0) memcpy(out + i * d, p + s * d, d * sizeof(float));
        else fill_f32(out + i * d, d, plane == 0 ? t->default_value : 0.0f);
        if (found) found[i] = s >= 0;
    }
}
void meo_assign_plane(meo_table* t, uint32_t plane, const int64_t* keys, const float* values, size_t n, uint8_t* found) {
    const uint32_t d = t->dim;
    float* p = plane_ptr(t, plane);
    for (size_t i = 0; i < n; ++i) {
        int64_t s = (!p || has_reserved(t, keys[i])) ? -1 : probe(t, keys[i], NULL);
        if (s >= 0) memcpy(p + s * d, values + i * d, d * sizeof(float));
        if (found) found[i] = s >= 0;
    }
}
void meo_remove(meo_table* t, const int64_t* keys, size_t n, uint8_t* found) {
    /* pass 1: found-mask as of before the call (duplicates all report the same) */
    if (found)
        for (size_t i = 0; i < n; ++i) found[i] = !reserved(keys[i]) && probe(t, keys[i], NULL) >= 0;
    for (size_t i = 0; i < n; ++i) {
        if (has_reserved(t, keys[i])) continue;
        int64_t s = probe(t, keys[i], NULL);
        if (s >= 0) { t->keys[s] = MEO_RECLAIMED_KEY; t->size--; }
    }
}
void meo_find_or_insert(meo_table* t, const int64_t* keys, size_t n, float* out, uint8_t* found) {
    const uint32_t d = t->dim;
    /* pass 1: found-mask as of before the call (so every occurrence of a new duplicate reports 0) */
    if (found)
        for (size_t i = 0; i < n; ++i) found[i] = !reserved(keys[i]) && probe(t, keys[i], NULL) >= 0;
    for (size_t i = 0; i < n; ++i) {
        if (has_reserved(t, keys[i])) { fill_f32(out + i * d, d, t->default_value); continue; }
        int64_t e, s = probe(t, keys[i], &e);
        if (s < 0) {
            s = place(t, keys[i], e);
            if (s >= 0) meo_initial_row(t, keys[i], t->values + s * d);
        }
        if (s >= 0) memcpy(out + i * d, t->values + s * d, d * sizeof(float));
        else fill_f32(out + i * d, d, t->default_value); /* table full */
    }
}
uint64_t meo_export(const meo_table* t, int64_t* keys_out, float* values_out, float* state1_out,
                    float* state2_out, uint64_t cap) {
    const uint32_t d = t->dim;
    uint64_t n = 0;
    for (uint64_t s = 0; s < t->capacity && n < cap; ++s) {
        if (t->keys[s] == MEO_EMPTY_KEY || t->keys[s] == MEO_RECLAIMED_KEY) continue;
        if (keys_out) keys_out[n] = t->keys[s];
        if (values_out) memcpy(values_out + n * d, t->values + s * d, d * sizeof(float));
        if (state1_out && t->s1) memcpy(state1_out + n * d, t->s1 + s * d, d * sizeof(float));
        if (state2_out && t->s2) memcpy(state2_out + n * d, t->s2 + s * d, d * sizeof(float));
        ++n;
    }
    return n;
}

/* ---- SPEC.md §4 ---------------------------------------------------------------------------------------- */
/* tiny open-addressing map key -> group index, for batch-local grouping */
struct gmap { uint64_t mask; int64_t* k; int64_t* v; };
static int gmap_init(struct gmap* g, size_t n) {
    uint64_t cap = 16; while (cap < 2 * n + 2) cap <<= 1;
    g->mask = cap - 1;
    g->k = (int64_t*)malloc(cap * sizeof(int64_t));
    g->v = (int64_t*)malloc(cap * sizeof(int64_t));
    if (!g->k || !g->v) return -1;
    for (uint64_t i = 0; i < cap; ++i) g->k[i] = MEO_EMPTY_KEY;
    return 0;
}
static void gmap_free(struct gmap* g) { free(g->k); free(g->v); }
/* returns pointer to value cell; *fresh=1 if the key was just added */
static int64_t* gmap_get(struct gmap* g, int64_t key, int* fresh) {
    uint64_t h = meo_mix64((uint64_t)key) & g->mask;
    while (g->k[h] != MEO_EMPTY_KEY && g->k[h] != key) h = (h + 1) & g->mask;
    *fresh = g->k[h] == MEO_EMPTY_KEY;
    g->k[h] = key;
    return &g->v[h];
}

uint64_t meo_dedup_sum(const int64_t* keys, const float* grads, size_t n, uint32_t dim, int64_t* uniq,
                       float* gsum, int64_t* inverse, uint32_t* counts) {
    struct gmap g;
    if (gmap_init(&g, n)) return 0;
    double* acc = (double*)calloc((n ? n : 1) * (size_t)dim, sizeof(double));
    uint32_t* cnt = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    uint64_t U = 0;
    for (size_t i = 0; i < n; ++i) {
        if (reserved(keys[i])) { if (inverse) inverse[i] = -1; continue; }
        int fresh; int64_t* cell = gmap_get(&g, keys[i], &fresh);
        if (fresh) { *cell = (int64_t)U; uniq[U] = keys[i]; ++U; }
        int64_t u = *cell;
        if (inverse) inverse[i] = u;
        cnt[u]++;
        if (grads) for (uint32_t j = 0; j < dim; ++j) acc[u * dim + j] += (double)grads[i * dim + j];
    }
    if (grads && gsum)
        for (uint64_t u = 0; u < U; ++u)
            for (uint32_t j = 0; j < dim; ++j) gsum[u * dim + j] = (float)acc[u * dim + j];
    /* single-occurrence keys: (float)(double)x == x, so "grad row unchanged" holds automatically */
    if (counts) memcpy(counts, cnt, U * sizeof(uint32_t));
    free(acc); free(cnt); gmap_free(&g);
    return U;
}

void meo_apply_adagrad(meo_table* t, const int64_t* keys, const float* grads, size_t n, float lr, float eps) {
    if (t->optimizer != MEO_OPT_ADAGRAD || n == 0) return;
    const uint32_t d = t->dim;
    int64_t* uniq = (int64_t*)malloc(n * sizeof(int64_t));
    float* gs = (float*)malloc(n * (size_t)d * sizeof(float));
    for (size_t i = 0; i < n; ++i) has_reserved(t, keys[i]);
    uint64_t U = meo_dedup_sum(keys, grads, n, d, uniq, gs, NULL, NULL);
    for (uint64_t u = 0; u < U; ++u) {
        int64_t s = probe(t, uniq[u], NULL);
        if (s < 0) continue;
        float* w = t->values + s * d; float* a = t->s1 + s * d; const float* g = gs + u * d;
        for (uint32_t j = 0; j < d; ++j) {
            float an = fmaf(g[j], g[j], a[j]);
            float q = g[j] / (sqrtf(an) + eps);
            w[j] = fmaf(-lr, q, w[j]);
            a[j] = an;
        }
    }
    free(uniq); free(gs);
}

void meo_apply_adam(meo_table* t, const int64_t* keys, const float* grads, size_t n, float lr, float beta1,
                    float beta2, float eps, uint64_t step) {
    if (t->optimizer != MEO_OPT_ADAM || n == 0) return;
    const uint32_t d = t->dim;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    int64_t* uniq = (int64_t*)malloc(n * sizeof(int64_t));
    float* gs = (float*)malloc(n * (size_t)d * sizeof(float));
    for (size_t i = 0; i < n; ++i) has_reserved(t, keys[i]);
    uint64_t U = meo_dedup_sum(keys, grads, n, d, uniq, gs, NULL, NULL);
    for (uint64_t u = 0; u < U; ++u) {
        int64_t s = probe(t, uniq[u], NULL);
        if (s < 0) continue;
        float* w = t->values + s * d; float* m = t->s1 + s * d; float* v = t->s2 + s * d;
        const float* g = gs + u * d;
        for (uint32_t j = 0; j < d; ++j) {
            float mn = fmaf(omb1, g[j] - m[j], m[j]);
            float gg = g[j] * g[j];
            float vn = fmaf(omb2, gg - v[j], v[j]);
            float q = mn / (sqrtf(vn) + eps);
            w[j] = fmaf(-step_size, q, w[j]);
            m[j] = mn; v[j] = vn;
        }
    }
    free(uniq); free(gs);
}

/* ---- SPEC.md §5 ---------------------------------------------------------------------------------------- */
void meo_partition(const int64_t* keys, size_t n, uint32_t n_shards, int64_t* send_keys, uint64_t* counts,
                   int64_t* perm) {
    uint64_t* off = (uint64_t*)calloc(n_shards + 1, sizeof(uint64_t));
    for (uint32_t p = 0; p < n_shards; ++p) counts[p] = 0;
    for (size_t i = 0; i < n; ++i) counts[meo_owner(keys[i], n_shards)]++;
    for (uint32_t p = 0; p < n_shards; ++p) off[p + 1] = off[p] + counts[p];
    for (size_t i = 0; i < n; ++i) {
        uint64_t q = off[meo_owner(keys[i], n_shards)]++;
        send_keys[q] = keys[i];
        perm[q] = (int64_t)i;
    }
    free(off);
}

// threads_test.cpp — the threading promise of the boundary (include/meepo_embedding.h: "concurrent mee_find* calls on one table from several
// host threads / streams are safe"; SURVEY.md §8b "One table handle may be used from multiple host threads"), from plain C++:
//   * 4 host threads, each with its own stream and its own buffers, issue 200 mee_find_ex calls each on ONE table — every thread with another
//     cache-policy flag set (per-call hints: nothing a concurrent caller reads is mutated) — and compare every result with a serial pass;
//   * one thread provokes an error in the middle (a null argument, then a flag combination that is refused): it must see ITS message in
//     mee_last_error(), while the other threads' error slots stay empty (the slot is thread-local).
// Uses only the C-ABI + the HIP runtime.  Exit code 0 = all checks passed.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "meepo_embedding.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define MEECK(x) do { int rc_ = (x); if (rc_ != MEE_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mee_last_error()); std::exit(3); } } while (0)

static uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
static float row_value(int64_t key, int j) { return (float)(mix64((uint64_t)key ^ mix64(2 + j)) >> 40) * 0x1p-24f - 0.5f; }

int main() {
    const uint32_t dim = 64;
    const size_t n_keys = 200000, batch = 8192;
    const int n_threads = 4, calls = 200;
    mee_config c{};
    c.struct_size = sizeof c; c.device = 0; c.capacity = (uint64_t)(n_keys / 0.75); c.dim = dim; c.max_batch = n_keys; c.default_value = -1.0f;
    mee_table* t = nullptr;
    MEECK(mee_table_create(&c, &t));
    std::vector<int64_t> keys(2 * n_keys);   // the second half is absent
    for (size_t i = 0; i < 2 * n_keys; ++i) keys[i] = (int64_t)mix64(1 + (i + 1) * 0x9E3779B97F4A7C15ull);
    std::vector<float> rows(n_keys * dim);
    for (size_t i = 0; i < n_keys; ++i) for (uint32_t j = 0; j < dim; ++j) rows[i * dim + j] = row_value(keys[i], (int)j);
    int64_t* d_all; float* d_rows;
    HIPCK(hipMalloc(&d_all, 2 * n_keys * 8)); HIPCK(hipMalloc(&d_rows, n_keys * dim * 4));
    HIPCK(hipMemcpy(d_all, keys.data(), 2 * n_keys * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_rows, rows.data(), n_keys * dim * 4, hipMemcpyHostToDevice));
    MEECK(mee_insert(t, d_all, d_rows, n_keys, nullptr));
    HIPCK(hipDeviceSynchronize());

    // every (thread, call) looks up its own window of the key array (present and absent keys mixed)
    auto window = [&](int th, int call) { return ((size_t)th * 7919 + (size_t)call * 104729) % (2 * n_keys - batch); };
    // serial pass: checksums of every call's result (sum of rows as integers bit patterns + found count), computed with plain mee_find
    std::vector<unsigned long long> want((size_t)n_threads * calls);
    {
        float* d_out; uint8_t* d_found;
        HIPCK(hipMalloc(&d_out, batch * dim * 4)); HIPCK(hipMalloc(&d_found, batch));
        std::vector<uint32_t> out(batch * dim); std::vector<uint8_t> found(batch);
        for (int th = 0; th < n_threads; ++th) for (int call = 0; call < calls; ++call) {
            MEECK(mee_find(t, d_all + window(th, call), batch, d_out, d_found, nullptr));
            HIPCK(hipMemcpy(out.data(), d_out, batch * dim * 4, hipMemcpyDeviceToHost));
            HIPCK(hipMemcpy(found.data(), d_found, batch, hipMemcpyDeviceToHost));
            unsigned long long h = 1469598103934665603ull;
            for (uint32_t v : out) h = (h ^ v) * 1099511628211ull;
            for (uint8_t v : found) h = (h ^ v) * 1099511628211ull;
            want[(size_t)th * calls + call] = h;
        }
        HIPCK(hipFree(d_out)); HIPCK(hipFree(d_found));
    }

    const uint32_t flags_of[4] = {MEE_FIND_DEFAULT, MEE_FIND_STREAM_STORES, MEE_FIND_STREAM_STORES | MEE_FIND_STREAM_ROWS, MEE_FIND_CACHED_STORES | MEE_FIND_STREAM_BUCKETS};
    std::atomic<int> failures{0};
    std::vector<std::thread> pool;
    for (int th = 0; th < n_threads; ++th) pool.emplace_back([&, th] {
        if (hipSetDevice(0) != hipSuccess) { ++failures; return; }
        hipStream_t st;
        if (hipStreamCreate(&st) != hipSuccess) { ++failures; return; }
        float* d_out[2]; uint8_t* d_found[2];   // two result buffers in rotation: a call's output is read back while the next call runs
        for (int b = 0; b < 2; ++b) if (hipMalloc(&d_out[b], batch * dim * 4) != hipSuccess || hipMalloc(&d_found[b], batch) != hipSuccess) { ++failures; return; }
        std::vector<uint32_t> out(batch * dim); std::vector<uint8_t> found(batch);
        if (strlen(mee_last_error()) != 0) { fprintf(stderr, "thread %d: error slot not empty at start: %s\n", th, mee_last_error()); ++failures; }
        for (int call = 0; call < calls; ++call) {
            const int b = call & 1;
            if (mee_find_ex(t, d_all + window(th, call), batch, d_out[b], d_found[b], flags_of[th], st) != MEE_OK) { fprintf(stderr, "thread %d call %d: %s\n", th, call, mee_last_error()); ++failures; break; }
            if (th == 2 && call == 100) {   // this thread alone provokes errors; its slot holds ITS messages
                if (mee_find_ex(t, nullptr, batch, d_out[b], d_found[b], 0, st) != MEE_ERR_INVALID_ARG || !strstr(mee_last_error(), "mee_find_ex: null argument")) { fprintf(stderr, "thread 2: expected its own null-argument error, got '%s'\n", mee_last_error()); ++failures; }
                if (mee_find_ex(t, d_all, batch, d_out[b], d_found[b], MEE_FIND_STREAM_STORES | MEE_FIND_CACHED_STORES, st) != MEE_ERR_INVALID_ARG || !strstr(mee_last_error(), "exclude each other")) { fprintf(stderr, "thread 2: expected the flag error, got '%s'\n", mee_last_error()); ++failures; }
                if (mee_find_ex(t, d_all, batch, d_out[b], d_found[b], 0x80u, st) != MEE_ERR_INVALID_ARG || !strstr(mee_last_error(), "unknown flag")) { fprintf(stderr, "thread 2: expected the unknown-flag error, got '%s'\n", mee_last_error()); ++failures; }
                if (mee_find_ex(t, d_all + window(th, call), batch, d_out[b], d_found[b], flags_of[th], st) != MEE_OK) ++failures;   // (the refused calls launched nothing: redo the lookup)
            }
            if (hipMemcpyAsync(out.data(), d_out[b], batch * dim * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(found.data(), d_found[b], batch, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { ++failures; break; }
            unsigned long long h = 1469598103934665603ull;
            for (uint32_t v : out) h = (h ^ v) * 1099511628211ull;
            for (uint8_t v : found) h = (h ^ v) * 1099511628211ull;
            if (h != want[(size_t)th * calls + call]) { fprintf(stderr, "thread %d call %d: result differs from the serial pass\n", th, call); ++failures; break; }
        }
        if (th != 2 && strlen(mee_last_error()) != 0) { fprintf(stderr, "thread %d sees another thread's error: %s\n", th, mee_last_error()); ++failures; }
        if (th == 2 && !strstr(mee_last_error(), "unknown flag")) { fprintf(stderr, "thread 2 lost its error message: '%s'\n", mee_last_error()); ++failures; }
        for (int b = 0; b < 2; ++b) { (void)hipFree(d_out[b]); (void)hipFree(d_found[b]); }
        (void)hipStreamDestroy(st);
    });
    for (auto& th : pool) th.join();
    MEECK(mee_table_destroy(t));
    if (failures.load()) { fprintf(stderr, "threads_test: %d failure(s)\n", failures.load()); return 1; }
    printf("threads_test ok: %d threads x %d mee_find_ex calls on one table, results equal to the serial pass, error slot thread-local\n", n_threads, calls);
    return 0;
}

// cabi_test.cpp — a plain C++ host program that uses ONLY include/meepo_embedding.h + the HIP runtime:
// evidence that the drop-in boundary is the C-ABI (no Python, no torch).  Exit code 0 = all checks passed.
// Checks use rows derived from the key, so no CPU table implementation is needed here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "meepo_embedding.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MEECK(x) do { int rc_ = (x); if (rc_ != MEE_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mee_last_error()); return 3; } } while (0)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 4; } } while (0)

static uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
static float row_value(int64_t key, int j, uint64_t seed) { return (float)(mix64((uint64_t)key ^ mix64(seed + j)) >> 40) * 0x1p-24f - 0.5f; }

int main() {
    const uint32_t dim = 64;
    const size_t n = 100000;
    CHECK(mee_abi_version() == MEE_ABI_VERSION);
    // bad config is rejected with a message, not a crash
    mee_config bad{};
    mee_table* t = nullptr;
    CHECK(mee_table_create(&bad, &t) == MEE_ERR_INVALID_ARG && strlen(mee_last_error()) > 0);

    mee_config c{};
    c.struct_size = sizeof c; c.device = 0; c.capacity = (uint64_t)(n / 0.75); c.dim = dim;
    c.optimizer = MEE_OPT_ADAGRAD; c.max_batch = n; c.default_value = -1.0f; c.initial_accumulator = 0.0f;
    MEECK(mee_table_create(&c, &t));
    mee_table_info info{};
    MEECK(mee_table_info_get(t, &info));
    CHECK(info.capacity % MEE_BUCKET_WIDTH == 0 && info.capacity >= c.capacity && info.dim == dim);

    std::vector<int64_t> keys(2 * n);
    for (size_t i = 0; i < 2 * n; ++i) keys[i] = (int64_t)mix64(1 + (i + 1) * 0x9E3779B97F4A7C15ull);
    std::vector<float> rows(n * dim), grads(n * dim);
    for (size_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < dim; ++j) { rows[i * dim + j] = row_value(keys[i], j, 2); grads[i * dim + j] = 0.02f * row_value(keys[i], j, 6); }

    hipStream_t st;
    HIPCK(hipStreamCreate(&st));
    int64_t* d_keys; float *d_rows, *d_grads, *d_out; uint8_t* d_found;
    HIPCK(hipMalloc(&d_keys, 2 * n * 8)); HIPCK(hipMalloc(&d_rows, n * dim * 4)); HIPCK(hipMalloc(&d_grads, n * dim * 4));
    HIPCK(hipMalloc(&d_out, 2 * n * dim * 4)); HIPCK(hipMalloc(&d_found, 2 * n));
    HIPCK(hipMemcpyAsync(d_keys, keys.data(), 2 * n * 8, hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_rows, rows.data(), n * dim * 4, hipMemcpyHostToDevice, st));
    HIPCK(hipMemcpyAsync(d_grads, grads.data(), n * dim * 4, hipMemcpyHostToDevice, st));

    MEECK(mee_insert(t, d_keys, d_rows, n, st));
    size_t sz = 0;
    MEECK(mee_size(t, &sz, st));
    CHECK(sz == n);
    MEECK(mee_find(t, d_keys, 2 * n, d_out, d_found, st));
    std::vector<float> out(2 * n * dim);
    std::vector<uint8_t> found(2 * n);
    HIPCK(hipMemcpyAsync(out.data(), d_out, 2 * n * dim * 4, hipMemcpyDeviceToHost, st));
    HIPCK(hipMemcpyAsync(found.data(), d_found, 2 * n, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    for (size_t i = 0; i < 2 * n; ++i) {
        CHECK(found[i] == (i < n));
        for (uint32_t j = 0; j < dim; ++j) CHECK(out[i * dim + j] == (i < n ? rows[i * dim + j] : -1.0f));
    }
    // one Adagrad step from acc = 0 (SPEC.md §4): acc' = g*g, w' = fma(-lr, g / (sqrt(acc') + eps), w)
    const float lr = 0.01f, eps = 1e-10f;
    MEECK(mee_apply_adagrad(t, d_keys, d_grads, n, lr, eps, st));
    MEECK(mee_find(t, d_keys, n, d_out, nullptr, st));
    HIPCK(hipMemcpyAsync(out.data(), d_out, n * dim * 4, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    for (size_t i = 0; i < n * dim; ++i) {
        const float g = grads[i], acc = fmaf(g, g, 0.0f);
        const float expect = fmaf(-lr, g / (sqrtf(acc) + eps), rows[i]);
        CHECK(out[i] == expect);
    }
    // export returns every pair once
    int64_t* d_ek; float* d_ev;
    HIPCK(hipMalloc(&d_ek, n * 8)); HIPCK(hipMalloc(&d_ev, n * dim * 4));
    size_t ne = 0;
    MEECK(mee_export(t, d_ek, d_ev, nullptr, nullptr, n, &ne, st));
    CHECK(ne == n);
    std::vector<int64_t> ek(n);
    HIPCK(hipMemcpy(ek.data(), d_ek, n * 8, hipMemcpyDeviceToHost));
    uint64_t xa = 0, xb = 0;
    for (size_t i = 0; i < n; ++i) { xa ^= mix64((uint64_t)ek[i]); xb ^= mix64((uint64_t)keys[i]); }
    CHECK(xa == xb);
    uint32_t status = 99;
    MEECK(mee_status(t, &status, st));
    CHECK(status == 0);
    // batch larger than max_batch is refused with a code, the table stays usable
    CHECK(mee_insert(t, d_keys, d_rows, n + 1, st) == MEE_ERR_BATCH_TOO_LARGE);
    MEECK(mee_table_destroy(t));
    printf("cabi_test ok: %zu keys inserted/found/updated/exported through the C-ABI\n", n);
    return 0;
}

// gather_ubench.hip — what can MI355X HBM sustain for the find access pattern?  (tuning evidence, not product)
//   per key: [optional B-byte random "bucket" read from a keys-sized array] + 256-B random row read + 256-B streaming write
// usage: gather_ubench <table_keys_M> <batch> <launches>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }

// MODE 0: rows only. 1: + 128-B line (16 lanes x 8 B). 2: + 64-B line (8 lanes x 8 B). 3: + 32-B (4 lanes)
template <int MODE, int R, bool NT_LOAD, int ROWL = 16>
__global__ __launch_bounds__(256) void gather_kernel(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values, uint64_t nslots,
                                                     const uint64_t* __restrict__ idx, uint64_t n, f32x4* __restrict__ out, int64_t* sink) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    int64_t acc = 0;
    for (uint64_t base = wave * 4 * R; base < n; base += nw * 4 * R) {
        uint64_t s[R]; int64_t kb[R]; f32x4 row[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; s[r] = i < n ? idx[i] : 0; }
        if (MODE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint64_t b = (s[r] >> 4) << 4;   // the 128-B line holding slot s
                const int lanes = MODE == 1 ? 16 : MODE == 2 ? 8 : 4;
                kb[r] = tl < lanes ? tkeys[b + tl] : 0;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {   // make the row address depend on the bucket data like a real probe does
                uint64_t m = __ballot(kb[r] == 0x7fffffffffffffffll);
                s[r] += (m >> (tile * 16)) & 1;   // always 0, but unknown to the compiler
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) if (tl < ROWL) row[r] = NT_LOAD ? __builtin_nontemporal_load(&values[s[r] * 16 + tl]) : values[s[r] * 16 + tl];
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; if (i < n && tl < ROWL) __builtin_nontemporal_store(row[r], &out[i * 16 + tl]); }
    }
    if (acc == 12345) *sink = acc;
}

__global__ void copy_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ b, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void fill_idx(uint64_t* idx, uint64_t n, uint64_t nslots, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        idx[i] = __umul64hi(mix64(i + seed * 0x9E3779B97F4A7C15ull), nslots);
}
__global__ void fill_f(float* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023);
}

template <typename F>
float time_us(F f, int launches) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) f(i);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < launches; ++i) f(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3f / launches < best) best = ms * 1e3f / launches;
    }
    return best;
}

int main(int argc, char** argv) {
    const uint64_t keysM = argc > 1 ? atoll(argv[1]) : 100;
    const uint64_t batch = argc > 2 ? atoll(argv[2]) : 262144;
    const int launches = argc > 3 ? atoi(argv[3]) : 200;
    const uint64_t nslots = ((uint64_t)(keysM * 1000000 / 0.75) + 15) / 16 * 16;
    int64_t* tkeys; f32x4* values; f32x4* out; uint64_t* idx; int64_t* sink;
    const int NB = 64;
    CK(hipMalloc(&tkeys, nslots * 8)); CK(hipMalloc(&values, nslots * 256)); CK(hipMalloc(&out, batch * 256));
    CK(hipMalloc(&idx, NB * batch * 8)); CK(hipMalloc(&sink, 8));
    fill_f<<<4096, 256>>>((float*)values, nslots * 64); fill_f<<<4096, 256>>>((float*)tkeys, nslots * 2);
    fill_idx<<<4096, 256>>>(idx, NB * batch, nslots, 7);
    CK(hipDeviceSynchronize());
    printf("table: %llu slots, rows %.1f GB, keys %.2f GB; batch %llu\n", (unsigned long long)nslots, nslots * 256 / 1e9, nslots * 8 / 1e9, (unsigned long long)batch);
    {   // streaming copy ceiling: 2 GB read + 2 GB write
        uint64_t n16 = (2ull << 30) / 16; if (n16 * 2 > nslots * 16) n16 = nslots * 8;
        float us = time_us([&](int) { copy_kernel<<<2048, 256>>>(values, values + n16, n16); }, 5);
        printf("%-44s %8.1f us  %7.1f GB/s (read+write)\n", "streaming copy 2GB->2GB", us, 2.0 * n16 * 16 / us / 1e3);
    }
#define RUNP(ROWL, label, bytes) { unsigned grid = (unsigned)((batch + 31) / 32); \
        float us = time_us([&](int i) { gather_kernel<0, 2, true, ROWL><<<grid, 256>>>(tkeys, values, nslots, idx + (uint64_t)(i % NB) * batch, batch, out, sink); }, launches); \
        printf("%-44s %8.2f us  %6.2f Gkeys/s  actual %7.1f GB/s\n", label, us, batch / us / 1e3, batch * (double)(bytes) / us / 1e3); }
    RUNP(16, "partial rows: 256B read+256B write", 520) RUNP(8, "partial rows: 128B read+128B write", 264) RUNP(4, "partial rows: 64B read+64B write", 136) RUNP(2, "partial rows: 32B read+32B write", 72)
#define RUN(MODE, R, NT, label, bytes) { unsigned grid = (unsigned)((batch + 16 * R - 1) / (16 * R)); \
        float us = time_us([&](int i) { gather_kernel<MODE, R, NT><<<grid, 256>>>(tkeys, values, nslots, idx + (uint64_t)(i % NB) * batch, batch, out, sink); }, launches); \
        printf("%-44s %8.2f us  %6.2f Gkeys/s  actual %7.1f GB/s  algorithmic(528) %7.1f GB/s = %.3f of 8TB/s\n", label, us, batch / us / 1e3, batch * (double)(bytes) / us / 1e3, batch * 528.0 / us / 1e3, batch * 528.0 / us / 1e3 / 8000); }
    RUN(0, 1, false, "rows only R=1", 520) RUN(0, 2, false, "rows only R=2", 520) RUN(0, 4, false, "rows only R=4", 520) RUN(0, 2, true, "rows only R=2 nt-load", 520)
    RUN(1, 2, false, "128B bucket + row R=2", 648) RUN(1, 4, false, "128B bucket + row R=4", 648) RUN(1, 2, true, "128B bucket + row R=2 nt-load", 648)
    RUN(2, 2, false, "64B bucket + row R=2", 584) RUN(2, 4, false, "64B bucket + row R=4", 584)
    RUN(3, 2, false, "32B bucket + row R=2", 552) RUN(3, 4, false, "32B bucket + row R=4", 552)
    return 0;
}

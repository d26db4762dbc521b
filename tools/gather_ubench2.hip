// gather_ubench2.hip — isolates what bounds a 256K-key random row gather on MI355X (tuning evidence, not product).
// usage: gather_ubench2 <table_keys_M> <batch>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }

template <int R, int ROWL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const f32x4* __restrict__ values, const uint64_t* __restrict__ idx, uint64_t n, f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * (BLOCK / 64);
    for (uint64_t base = wave * 4 * R; base < n; base += nw * 4 * R) {
        uint64_t s[R]; f32x4 row[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; s[r] = i < n ? idx[i] : 0; }
#pragma unroll
        for (int r = 0; r < R; ++r) if (tl < ROWL) row[r] = __builtin_nontemporal_load(&values[s[r] * 16 + tl]);
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; if (i < n && tl < ROWL) __builtin_nontemporal_store(row[r], &out[i * 16 + tl]); }
    }
}
// software-pipelined persistent variant: rows of round i+1 are requested before round i is stored
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather_pipe_kernel(const f32x4* __restrict__ values, const uint64_t* __restrict__ idx, uint64_t n, f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * (BLOCK / 64);
    const uint64_t stride = nw * 4 * R;
    uint64_t base = wave * 4 * R;
    if (base >= n) return;
    uint64_t s[R], s2[R]; f32x4 row[R], row2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; s[r] = i < n ? idx[i] : 0; }
#pragma unroll
    for (int r = 0; r < R; ++r) row[r] = __builtin_nontemporal_load(&values[s[r] * 16 + tl]);
#pragma unroll
    for (int r = 0; r < R; ++r) { uint64_t i = base + stride + r * 4 + tile; s2[r] = i < n ? idx[i] : 0; }
    while (true) {
        const uint64_t nxt = base + stride;
        const bool more = nxt < n;   // wave-uniform
        if (more) {
#pragma unroll
            for (int r = 0; r < R; ++r) row2[r] = __builtin_nontemporal_load(&values[s2[r] * 16 + tl]);
#pragma unroll
            for (int r = 0; r < R; ++r) { uint64_t i = nxt + stride + r * 4 + tile; s2[r] = i < n ? idx[i] : 0; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; if (i < n) __builtin_nontemporal_store(row[r], &out[i * 16 + tl]); }
        if (!more) break;
#pragma unroll
        for (int r = 0; r < R; ++r) row[r] = row2[r];
        base = nxt;
    }
}
__global__ void fill_idx(uint64_t* idx, uint64_t n, uint64_t nslots, uint64_t seed, int mode) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        idx[i] = mode == 0 ? __umul64hi(mix64(i + seed * 0x9E3779B97F4A7C15ull), nslots) : mode == 1 ? (i % nslots) : __umul64hi(mix64(i + seed), 262144);
}
__global__ void fill_f(float* p, uint64_t n) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023); }
template <typename F> float time_us(F f, int launches) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) f(i);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); for (int i = 0; i < launches; ++i) f(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms * 1e3f / launches < best) best = ms * 1e3f / launches;
    }
    return best;
}
int main(int argc, char** argv) {
    const uint64_t keysM = argc > 1 ? atoll(argv[1]) : 100, batch = argc > 2 ? atoll(argv[2]) : 262144;
    const int launches = 200, NB = 32;
    const uint64_t nslots = ((uint64_t)(keysM * 1000000 / 0.75) + 15) / 16 * 16;
    f32x4 *values, *out; uint64_t* idx[3];
    CK(hipMalloc(&values, nslots * 256)); CK(hipMalloc(&out, batch * 256));
    fill_f<<<4096, 256>>>((float*)values, nslots * 64);
    for (int m = 0; m < 3; ++m) { CK(hipMalloc(&idx[m], NB * batch * 8)); fill_idx<<<4096, 256>>>(idx[m], NB * batch, nslots, 7, m); }
    CK(hipDeviceSynchronize());
    printf("table %llu slots (%.1f GB), batch %llu\n", (unsigned long long)nslots, nslots * 256 / 1e9, (unsigned long long)batch);
    const char* pat[3] = {"random", "sequential", "random-in-64MB"};
#define RUN(R, ROWL, BLOCK, GRIDCAP) for (int m = 0; m < 3; ++m) { size_t g = (batch + 4 * R * (BLOCK / 64) - 1) / (4 * R * (BLOCK / 64)); if (GRIDCAP && g > GRIDCAP) g = GRIDCAP; \
        float us = time_us([&](int i) { gather_kernel<R, ROWL, BLOCK><<<(unsigned)g, BLOCK>>>(values, idx[m] + (uint64_t)(i % NB) * batch, batch, out); }, launches); \
        printf("plain R=%-2d rowB=%-3d block=%-4d grid=%-6zu %-15s %7.2f us %6.2f Gkeys/s %7.1f GB/s\n", R, ROWL * 16, BLOCK, g, pat[m], us, batch / us / 1e3, batch * (8.0 + ROWL * 32) / us / 1e3); }
#define RUNP(R, BLOCK, GRID) for (int m = 0; m < 3; ++m) { size_t g = GRID; \
        float us = time_us([&](int i) { gather_pipe_kernel<R, BLOCK><<<(unsigned)g, BLOCK>>>(values, idx[m] + (uint64_t)(i % NB) * batch, batch, out); }, launches); \
        printf("pipe  R=%-2d rowB=256 block=%-4d grid=%-6zu %-15s %7.2f us %6.2f Gkeys/s %7.1f GB/s\n", R, BLOCK, g, pat[m], us, batch / us / 1e3, batch * 520.0 / us / 1e3); }
    RUN(1, 2, 256, 0) RUN(4, 2, 256, 0) RUN(16, 2, 256, 0)
    RUN(1, 16, 256, 0) RUN(2, 16, 256, 0) RUN(8, 16, 256, 0) RUN(2, 16, 1024, 0) RUN(2, 16, 64, 0) RUN(2, 16, 256, 2048)
    RUNP(1, 256, 2048) RUNP(2, 256, 2048) RUNP(2, 256, 1024) RUNP(4, 256, 1024) RUNP(2, 512, 1024) RUNP(1, 256, 4096) RUNP(4, 256, 512)
    return 0;
}

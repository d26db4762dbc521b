// gather_ubench3.hip — does co-locating a bucket's key line with its 16 rows (one 4224-B record per bucket) beat
// separate key / value arrays?  Same dependent chain (index -> 128-B bucket line -> 256-B row -> 256-B store).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }

// LAYOUT 0: keys[nslots] + values[nslots][16 f32x4].  LAYOUT 1: records of 264 f32x4: [8 f32x4 of keys][16 rows x 16 f32x4]
template <int LAYOUT, int R>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tkeys, const f32x4* __restrict__ values, const uint64_t* __restrict__ idx, uint64_t n, f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    for (uint64_t base = wave * 4 * R; base < n; base += nw * 4 * R) {
        uint64_t s[R]; int64_t kb[R]; f32x4 row[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; s[r] = i < n ? idx[i] : 0; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t b = s[r] >> 4;
            kb[r] = LAYOUT == 0 ? tkeys[b * 16 + tl] : reinterpret_cast<const int64_t*>(values + b * 264)[tl];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t m = __ballot(kb[r] == 0x7fffffffffffffffll); s[r] += (m >> (tile * 16)) & 1; }
#pragma unroll
        for (int r = 0; r < R; ++r) row[r] = LAYOUT == 0 ? values[s[r] * 16 + tl] : values[(s[r] >> 4) * 264 + 8 + (s[r] & 15) * 16 + tl];
#pragma unroll
        for (int r = 0; r < R; ++r) { uint64_t i = base + r * 4 + tile; if (i < n) out[i * 16 + tl] = row[r]; }
    }
}
__global__ void fill_idx(uint64_t* idx, uint64_t n, uint64_t nslots, uint64_t seed) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) idx[i] = __umul64hi(mix64(i + seed * 0x9E3779B97F4A7C15ull), nslots); }
__global__ void fill_f(float* p, uint64_t n) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023); }
template <typename F> float time_us(F f, int launches) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) f(i);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); for (int i = 0; i < launches; ++i) f(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms * 1e3f / launches < best) best = ms * 1e3f / launches; }
    return best;
}
int main(int argc, char** argv) {
    const uint64_t keysM = argc > 1 ? atoll(argv[1]) : 100, batch = argc > 2 ? atoll(argv[2]) : 262144;
    const uint64_t nslots = ((uint64_t)(keysM * 1000000 / 0.75) + 15) / 16 * 16, nb = nslots / 16;
    int64_t* tkeys; f32x4 *values, *rec, *out; uint64_t* idx; const int NB = 32;
    CK(hipMalloc(&tkeys, nslots * 8)); CK(hipMalloc(&values, nslots * 256)); CK(hipMalloc(&rec, nb * 264 * 16)); CK(hipMalloc(&out, batch * 256)); CK(hipMalloc(&idx, NB * batch * 8));
    fill_f<<<4096, 256>>>((float*)values, nslots * 64); fill_f<<<4096, 256>>>((float*)tkeys, nslots * 2); fill_f<<<4096, 256>>>((float*)rec, nb * 264 * 4);
    fill_idx<<<4096, 256>>>(idx, NB * batch, nslots, 7); CK(hipDeviceSynchronize());
    printf("%llu slots; separate arrays %.1f GB, records %.1f GB; batch %llu\n", (unsigned long long)nslots, nslots * 264 / 1e9, nb * 264 * 16 / 1e9, (unsigned long long)batch);
    for (int rep = 0; rep < 2; ++rep) {
        unsigned g2 = (unsigned)((batch + 31) / 32);
        float a = time_us([&](int i) { k<0, 2><<<g2, 256>>>(tkeys, values, idx + (uint64_t)(i % NB) * batch, batch, out); }, 200);
        float b = time_us([&](int i) { k<1, 2><<<g2, 256>>>(tkeys, rec, idx + (uint64_t)(i % NB) * batch, batch, out); }, 200);
        printf("R=2: separate key/value arrays %7.2f us | bucket records (keys + rows contiguous) %7.2f us  (%.1f %%)\n", a, b, 100.0 * (a - b) / a);
    }
    return 0;
}

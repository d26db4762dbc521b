// rmw_layout_ubench.hip — the apply's main pass as a bare access pattern: per key, read a 256-B grad row (sequential), read-modify-write
// a 256-B value row and a 256-B accumulator row at a random slot.  (a) two separate planes (today's layout), (b) one interleaved
// record of 512 B per slot.  Does the interleaved layout buy anything?   build: hipcc --offload-arch=gfx950 -O3 … -o build/rmw_layout_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }

template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void rmw(float4* __restrict__ v, float4* __restrict__ a, const float4* __restrict__ grads, uint64_t n_slots, uint32_t n, uint64_t seed) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint32_t i = base + tile;
        if (i >= n) continue;
        const uint64_t slot = mix64(i ^ seed) % n_slots;
        const float4 g = grads[(uint64_t)i * 16 + tl];
        float4 *pv, *pa;
        if (INTERLEAVED) { pv = v + slot * 32 + tl; pa = pv + 16; }
        else { pv = v + slot * 16 + tl; pa = a + slot * 16 + tl; }
        float4 w = *pv, x = *pa;
        x.x += g.x * g.x; x.y += g.y * g.y; x.z += g.z * g.z; x.w += g.w * g.w;
        w.x -= 0.01f * g.x / (sqrtf(x.x) + 1e-10f); w.y -= 0.01f * g.y / (sqrtf(x.y) + 1e-10f);
        w.z -= 0.01f * g.z / (sqrtf(x.z) + 1e-10f); w.w -= 0.01f * g.w / (sqrtf(x.w) + 1e-10f);
        *pv = w; *pa = x;
    }
}

int main() {
    const uint64_t n_slots = 133333808ull;   // the 100M-key table of configs[1]/[2]
    const uint32_t n = 1u << 18;
    float4 *v, *a, *g;
    if (hipMalloc(&v, n_slots * 512) != hipSuccess || hipMalloc(&g, (size_t)n * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    a = v + n_slots * 16;   // separate planes: second half of the same allocation
    hipMemset(v, 0, n_slots * 512); hipMemset(g, 0, (size_t)n * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 50; ++k) {
                if (mode == 0) rmw<false><<<n / 16, 256>>>(v, a, g, n_slots, n, 77 + k + rep * 100);
                else rmw<true><<<n / 16, 256>>>(v, a, g, n_slots, n, 77 + k + rep * 100);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per 256K keys (%.2f TB/s of 1280 B/key)\n", mode ? "interleaved 512-B records" : "two separate planes      ", ms * 20, n * 1280.0 / (ms * 20) / 1e6);
        }
    return 0;
}

// anyorder_ubench.hip — does hipExtAnyOrderLaunch let consecutive launches on ONE stream overlap on gfx950 (hip_ext.h says "not supported on GFX9xx")?
// A latency-floor-bound kernel (dependent chain of 4 random loads per lane, small grid) launched 200 times back to back, with and without the flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void chase(const uint32_t* __restrict__ buf, uint32_t mask, uint32_t* out, uint32_t seed) {
    uint32_t i = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    uint32_t v = buf[i & mask];
    v = buf[(v + i) & mask]; v = buf[(v ^ i) & mask]; v = buf[(v + 7 * i) & mask];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    const uint32_t n = 1u << 28;   // 1 GiB of uint32: far beyond the caches
    uint32_t *buf, *out;
    hipMalloc(&buf, (size_t)n * 4); hipMalloc(&out, 1u << 22);
    hipMemset(buf, 0x5a, (size_t)n * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            for (int k = 0; k < 200; ++k) {
                if (mode == 0) chase<<<1024, 256, 0, st>>>(buf, n - 1, out, k);
                else hipExtLaunchKernelGGL(chase, dim3(1024), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, buf, n - 1, out, (uint32_t)k);
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.2f us per launch (%s)\n", mode ? "hipExtAnyOrderLaunch" : "in-order launch    ", ms * 1e3 / 200, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}

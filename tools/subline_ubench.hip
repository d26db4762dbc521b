// subline_ubench.hip — does gfx950 fetch less than a 128-B line from HBM for a narrow random read, under ANY load flavour?
// N random reads over a 4 GiB buffer (far beyond L2 + Infinity Cache); each 16-lane tile reads one aligned chunk of W bytes
// (W = 128: all 16 lanes x 8 B; 64: 8 lanes; 32: 4 lanes) with a given cache policy; accesses/us tells the granularity.
// build: hipcc --offload-arch=gfx950 -O3 tools/subline_ubench.hip -o build/subline_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }

template <int POLICY>
__device__ __forceinline__ int64_t ld(const int64_t* p) {
    if constexpr (POLICY == 0) return *p;
    else if constexpr (POLICY == 1) return __builtin_nontemporal_load(p);
    else if constexpr (POLICY == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int LANES, int POLICY>
__global__ __launch_bounds__(256) void probe(const int64_t* __restrict__ buf, uint64_t n_lines, uint64_t n, uint64_t seed, int64_t* sink) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    int64_t acc = 0;
    for (uint64_t base = wave * 8; base < n; base += n_waves * 8) {
        int64_t v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint64_t i = base + r * 4 + tile;
            const uint64_t line = mix64(i ^ seed) % n_lines;
            v[r] = tl < LANES ? ld<POLICY>(buf + line * 16 + tl) : 0;
        }
        acc += v[0] ^ v[1];
    }
    if (acc == 0x1234567) *sink = acc;
}

int main() {
    const uint64_t bytes = 4ull << 30, n_lines = bytes / 128, n = 1ull << 22;
    int64_t *buf, *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) kern<<<(unsigned)(n / 32), 256>>>(buf, n_lines, n, 1234 + rep * 10 + k, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 10 < best) best = ms / 10;
        }
        printf("%-34s %8.1f us per %llu reads -> %7.1f reads/ns\n", name, best * 1e3, (unsigned long long)n, n / (best * 1e6));
    };
    run(probe<16, 0>, "128 B, plain");
    run(probe<8, 0>, " 64 B, plain");
    run(probe<4, 0>, " 32 B, plain");
    run(probe<16, 1>, "128 B, nontemporal");
    run(probe<8, 1>, " 64 B, nontemporal");
    run(probe<4, 1>, " 32 B, nontemporal");
    run(probe<16, 2>, "128 B, agent-scope (sc1)");
    run(probe<8, 2>, " 64 B, agent-scope (sc1)");
    run(probe<4, 2>, " 32 B, agent-scope (sc1)");
    run(probe<16, 3>, "128 B, system-scope (sc0 sc1)");
    run(probe<8, 3>, " 64 B, system-scope (sc0 sc1)");
    run(probe<4, 3>, " 32 B, system-scope (sc0 sc1)");
    return 0;
}

// meepo_device.h — device-side building blocks shared by the gfx950 kernels (wave64 only).
//
// Geometry used everywhere: a wave (64 lanes) is split into 4 "tiles" of 16 lanes.  One tile serves one key:
// its 16 lanes load the 16 keys of a bucket (one 128-byte line, 8 B per lane), a wave-wide __ballot turns the
// compares into a 64-bit mask of which each tile reads its own 16 bits, and the same 16 lanes then move the
// embedding row as float4 (16 lanes x 16 B = 256 B per instruction for dim 64).
//
// Reference anchor: /root/reference/README.md:2 (no code upstream); semantics: SPEC.md §1-§4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mee {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // what __builtin_nontemporal_load / _store move as one dwordx4

constexpr int64_t kEmpty = INT64_MIN;
constexpr int64_t kReclaimed = INT64_MIN + 1;
constexpr int kW = 16;                                   // bucket width == tile width
constexpr unsigned long long kBias = 0x8000000000000000ull;  // scratch stores key^kBias so that 0 == empty
constexpr uint32_t kNoGroup = 0xFFFFFFFFu;
// Slot handles of mee_find_located / mee_find_or_insert_located: bits [0, 40) = the slot, bits [40, 62) = the table's layout epoch when the
// handle was made (bumped by remove / clear / reserve: whatever can move or free a row).  mee_apply_*_located ignores a handle of another
// epoch and raises MEE_STATUS_STALE_HANDLE instead of updating whatever lives in that slot now.  -1 = absent.
constexpr int kHandleSlotBits = 40;
constexpr int64_t kHandleSlotMask = (1ll << kHandleSlotBits) - 1;
constexpr uint32_t kHandleEpochMask = (1u << 22) - 1;
// decode: the slot a handle names, or -1 (absent, out of range, or made under another layout epoch: `stale`)
__device__ __forceinline__ int64_t handle_slot(int64_t h, int64_t tag, uint64_t capacity, bool& stale) {
    stale = h >= 0 && (h & ~kHandleSlotMask) != tag;
    const int64_t s = h & kHandleSlotMask;
    return (h >= 0 && !stale && (uint64_t)s < capacity) ? s : -1;   // a handle is the caller's data: never index past the planes
}

// SPEC.md §1
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31; return x;
}
__host__ __device__ __forceinline__ uint64_t mix64b(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33; return x;
}
__device__ __forceinline__ uint64_t bucket_of(int64_t key, uint64_t nb) { return __umul64hi(mix64((uint64_t)key), nb); }
// SPEC.md §1/§2: stride (in buckets) of a key's probe sequence — double hashing, 1 <= stride < nb
__device__ __forceinline__ uint64_t step_of(int64_t key, uint64_t nb) { return nb > 1 ? 1 + __umul64hi(mix64b((uint64_t)key), nb - 1) : 1; }
__device__ __forceinline__ uint64_t next_bucket(uint64_t b, uint64_t stride, uint64_t nb) { b += stride; return b >= nb ? b - nb : b; }
__device__ __forceinline__ uint32_t owner_of(int64_t key, uint32_t g) { return (uint32_t)__umul64hi(mix64b((uint64_t)key), (uint64_t)g); }
__device__ __forceinline__ bool reserved_key(int64_t k) { return k <= kReclaimed; }

// Table keys are read with plain loads by read-only kernels (find/assign/apply: no key changes while they run)
// and with agent-scope relaxed atomic loads (global_load … sc1: bypasses the per-CU L1) by kernels that claim
// slots concurrently, so a retry after a lost CAS never re-reads a stale L1 line.
template <bool COHERENT>
__device__ __forceinline__ int64_t load_table_key(const int64_t* p) {
    if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

__device__ __forceinline__ uint32_t tile_bits(uint64_t wave_mask, int tile) {
    return (uint32_t)(wave_mask >> (tile * kW)) & 0xFFFFu;
}

// Locate `key` (SPEC.md §2 probe sequence); with CLAIM also place it when absent: CAS into the first RECLAIMED slot
// met on the way if there is one, else into the first EMPTY slot of the bucket that ended the probe.  Must be called
// by ALL 64 lanes in convergent control flow; tiles without work pass active=false.  Returns the slot or -1.
template <bool CLAIM, bool COHERENT>
__device__ __forceinline__ int64_t tile_locate(int64_t* __restrict__ tkeys, uint64_t nb, int64_t key, bool active,
                                               int tile, int tl, bool& is_new, bool& full) {
    const uint64_t b0 = bucket_of(key, nb), stride = step_of(key, nb);
    uint64_t b = b0;
    uint64_t steps = 0;
    uint32_t retries = 0;  // lost-CAS re-reads; bounded so that every wave reaches the exit
    int64_t tomb = -1;     // first RECLAIMED slot seen on this probe
    bool pend = active;
    int64_t slot = -1;
    is_new = false; full = false;
    while (__any(pend)) {
        const int64_t k = pend ? load_table_key<COHERENT>(tkeys + b * kW + tl) : kEmpty;
        const uint32_t tm = tile_bits(__ballot(pend && k == key), tile);
        const uint32_t te = tile_bits(__ballot(pend && k == kEmpty), tile);
        long long cas_old = 0;
        int64_t target = -1;
        long long expect = kEmpty;
        if constexpr (CLAIM) {
            const uint32_t tr = tile_bits(__ballot(pend && k == kReclaimed), tile);
            if (pend && tomb < 0 && tr) tomb = (int64_t)(b * kW) + (__ffs(tr) - 1);
            const bool last = steps + 1 >= nb;  // the whole sequence has been scanned after this bucket
            if (pend && !tm && (te || (last && tomb >= 0))) {
                target = tomb >= 0 ? tomb : (int64_t)(b * kW) + (__ffs(te) - 1);
                expect = tomb >= 0 ? kReclaimed : kEmpty;
                if (tl == 0)
                    cas_old = (long long)atomicCAS((unsigned long long*)(tkeys + target), (unsigned long long)expect,
                                                   (unsigned long long)key);
            }
            cas_old = __shfl(cas_old, tile * kW);
        }
        if (pend) {
            if (tm) { slot = (int64_t)(b * kW) + (__ffs(tm) - 1); pend = false; }
            else if (CLAIM && target >= 0) {
                if (cas_old == expect) { slot = target; is_new = true; pend = false; }
                else if (cas_old == key) { slot = target; pend = false; }
                else if (++retries > (1u << 20)) { full = true; pend = false; }
                else if (expect == kReclaimed) { tomb = -1; b = b0; steps = 0; }  // lost a tombstone: start over
                // else: another key took that EMPTY slot — re-read the same bucket
            } else if (te) pend = false;  // absent
            else if (++steps >= nb) { full = true; pend = false; }
            else b = next_bucket(b, stride, nb);
        }
    }
    return slot;
}

// SPEC.md §3 "Initial row", four consecutive elements starting at j0
__device__ __forceinline__ float4 initial_row4(int64_t key, uint32_t j0, uint32_t initializer, float init_scale,
                                               uint64_t init_seed, float default_value) {
    if (initializer == 0) return make_float4(default_value, default_value, default_value, default_value);
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t h = mix64((uint64_t)key ^ mix64(init_seed + j0 + q));
        const float u = (float)(h >> 40) * 0x1p-24f;
        r[q] = init_scale * (2.0f * u - 1.0f);
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// SPEC.md §4 (explicit fma only where the spec writes it; file is compiled with -ffp-contract=off)
__device__ __forceinline__ void adagrad1(float& w, float& a, float g, float lr, float eps) {
    const float an = __builtin_fmaf(g, g, a);
    const float q = g / (__builtin_sqrtf(an) + eps);
    w = __builtin_fmaf(-lr, q, w);
    a = an;
}
__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, float step_size, float omb1, float omb2,
                                      float eps) {
    const float mn = __builtin_fmaf(omb1, g - m, m);
    const float gg = g * g;
    const float vn = __builtin_fmaf(omb2, gg - v, v);
    const float q = mn / (__builtin_sqrtf(vn) + eps);
    w = __builtin_fmaf(-step_size, q, w);
    m = mn; v = vn;
}

}  // namespace mee

// meepo_export.hip — everything that walks the whole table or reports on it: export, size, rehash (mee_reserve), the hit-counter scan of
// the hot/cold policy, probe-length statistics, the status word; their half of the C-ABI.  Table layout: meepo_table.hip.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "meepo_apply_part.h"

namespace mee {

// ---- access statistics for the hot/cold policy: keys whose hit counter lies in [lo, hi], optional reset ---------------
__global__ __launch_bounds__(256) void hits_scan_kernel(const int64_t* __restrict__ tkeys, uint32_t* hits, uint64_t capacity, uint32_t lo,
                                                        uint32_t hi, int reset, int64_t* keys_out, uint64_t cap, OpCounters* op) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    constexpr uint64_t kSpan = 64ull * 16;
    for (uint64_t c0 = wave * kSpan; c0 < capacity; c0 += n_waves * kSpan) {
        int64_t k[16];
        uint64_t m[16];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t s = c0 + (uint64_t)j * 64 + lane;
            k[j] = s < capacity ? tkeys[s] : kEmpty;
            const uint32_t hcount = s < capacity ? hits[s] : 0;
            if (reset && s < capacity && hcount) hits[s] = 0;
            m[j] = __ballot(!reserved_key(k[j]) && hcount >= lo && hcount <= hi);
            total += (uint32_t)__popcll(m[j]);
        }
        if (!total) continue;  // wave-uniform
        unsigned long long pos0 = 0;
        if (lane == 0) pos0 = atomicAdd(&op->n_export, (unsigned long long)total);
        pos0 = __shfl(pos0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((m[j] >> lane) & 1) {
                const uint64_t pos = pos0 + (uint64_t)__popcll(m[j] & ((1ull << lane) - 1));
                if (pos < cap) keys_out[pos] = k[j];
            }
            pos0 += (uint64_t)__popcll(m[j]);
        }
    }
}

// ---- probe-length statistics (SURVEY.md §8d "mean probe length"): buckets visited per lookup, summed over the batch ----
__global__ __launch_bounds__(256) void probe_length_kernel(const int64_t* __restrict__ tkeys, uint64_t nb, const int64_t* __restrict__ keys,
                                                           uint64_t n, OpCounters* op) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    unsigned long long visited = 0, hist = 0;   // hist: four 16-bit counters (a wave takes at most 4096 keys per counter before they are flushed below)
    uint32_t rounds = 0;
    auto flush = [&]() {
        unsigned long long h = hist;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {   // field-wise sums: 64 lanes x 2^10 per field stay below 2^16
            const unsigned long long o = (unsigned long long)(uint32_t)__shfl_down((int)(uint32_t)h, d) | (unsigned long long)(uint32_t)__shfl_down((int)(uint32_t)(h >> 32), d) << 32;
            h += o;
        }
        if (lane == 0)
            for (int q = 0; q < 4; ++q) { const unsigned long long c = (h >> (16 * q)) & 0xFFFFull; if (c) atomicAdd(&op->hist[q], c); }
        hist = 0;
    };
    for (uint64_t base = wave * 4; base < n; base += n_waves * 4) {
        const uint64_t i = base + tile;
        const int64_t key = i < n ? keys[i] : kEmpty;
        bool pend = i < n && !reserved_key(key);
        uint64_t b = bucket_of(key, nb), steps = 0;
        uint32_t mine = 0;
        while (__any(pend)) {
            const int64_t k = pend ? tkeys[b * kW + tl] : kEmpty;
            const uint32_t tm = tile_bits(__ballot(pend && k == key), tile);
            const uint32_t te = tile_bits(__ballot(pend && k == kEmpty), tile);
            if (pend) {
                if (tl == 0) { ++visited; ++mine; }
                if (tm || te || ++steps >= nb) pend = false;
                else b = next_bucket(b, step_of(key, nb), nb);
            }
        }
        if (mine) hist += 1ull << (16 * (min(mine, 4u) - 1u));
        if (++rounds == 1000) { flush(); rounds = 0; }   // (wave-uniform)
    }
    flush();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) visited += __shfl_down(visited, d);
    if (lane == 0 && visited) atomicAdd(&op->n_export, visited);
}

// ---- size (SPEC.md §3): count stored keys by scanning the key plane (keeps every atomic off the insert path) ----
__global__ __launch_bounds__(256) void count_kernel(const int64_t* __restrict__ tkeys, uint64_t capacity, OpCounters* op) {
    __shared__ uint32_t wsum[4];
    uint32_t c = 0;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < capacity; s += (uint64_t)gridDim.x * blockDim.x)
        c += !reserved_key(tkeys[s]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (tot) atomicAdd(&op->n_export, (unsigned long long)tot);
    }
}

// ---- reserve / rehash (SPEC.md §3): every stored pair moves, device to device, into key/row planes of another capacity ----
// A wave reads 64 consecutive old slots (one coalesced key load), then its four tiles take the stored keys four at a
// time: claim a slot in the new key plane (same CAS protocol as insert: all keys are distinct, claims of different
// waves race for slots) and copy the row, the optimizer planes and the hit counter.
__global__ __launch_bounds__(256) void rehash_kernel(const int64_t* __restrict__ okeys, const float4* __restrict__ ov,
                                                     const float4* __restrict__ o1, const float4* __restrict__ o2,
                                                     const uint32_t* __restrict__ ohits, uint64_t old_capacity, int64_t* nkeys,
                                                     float4* nv, float4* n1, float4* n2, uint32_t* nhits, uint64_t nnb,
                                                     uint32_t dim4, Counters* ctr) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t c0 = wave * 64; c0 < old_capacity; c0 += n_waves * 64) {
        const uint64_t s = c0 + lane;
        const int64_t k = s < old_capacity ? okeys[s] : kEmpty;
        uint64_t rest = __ballot(!reserved_key(k));
        while (rest) {  // wave-uniform
            uint64_t mm = rest;
            int p = -1;
            for (int q = 0; q <= tile; ++q) {
                if (mm) { p = __ffsll((unsigned long long)mm) - 1; mm &= mm - 1; } else p = -1;
            }
            const int64_t key = __shfl(k, p >= 0 ? p : 0);
            bool is_new, full;
            const int64_t slot = tile_locate<true, true>(nkeys, nnb, key, p >= 0, tile, tl, is_new, full);
            if (p >= 0) {
                if (slot >= 0) {
                    const uint64_t src = (c0 + (uint64_t)p) * dim4, dst = (uint64_t)slot * dim4;
                    for (uint32_t c = tl; c < dim4; c += 16) {
                        nv[dst + c] = ov[src + c];
                        if (n1) n1[dst + c] = o1[src + c];
                        if (n2) n2[dst + c] = o2[src + c];
                    }
                    if (nhits && tl == 0) nhits[slot] = ohits[c0 + p];
                } else if (tl == 0) {
                    atomicOr(&ctr->status, (uint32_t)MEE_STATUS_TABLE_FULL);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rest &= rest - 1;
        }
    }
}

// ---- export (SPEC.md §3) -----------------------------------------------------------------------------------
// A wave owns a chunk of 1024 consecutive slots.  Pass A: 16 coalesced key loads, ballot + popcount -> occupied
// count, ONE atomic reserves the chunk's output range.  Pass B: per 64-slot group, ranks from the ballot mask;
// keys are written by their own lane, rows are copied four at a time (one per 16-lane tile).
constexpr int kExportGroups = 16;

__global__ __launch_bounds__(256) void export_kernel(const int64_t* __restrict__ tkeys, const float4* __restrict__ values,
                                                     const float4* __restrict__ s1, const float4* __restrict__ s2,
                                                     uint64_t begin, uint64_t capacity /* = end of the slot range */, uint32_t dim4,
                                                     int64_t* keys_out, float4* values_out,
                                                     float4* s1_out, float4* s2_out, uint64_t cap, OpCounters* op) {
    const int lane = threadIdx.x & 63, tile = lane >> 4, tl = lane & 15;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    constexpr uint64_t kSpan = 64ull * kExportGroups;
    for (uint64_t c0 = begin + wave * kSpan; c0 < capacity; c0 += n_waves * kSpan) {
        int64_t k[kExportGroups];
        uint64_t m[kExportGroups];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < kExportGroups; ++j) {
            const uint64_t s = c0 + (uint64_t)j * 64 + lane;
            k[j] = s < capacity ? tkeys[s] : kEmpty;
            m[j] = __ballot(!reserved_key(k[j]));
            total += (uint32_t)__popcll(m[j]);
        }
        if (!total) continue;  // wave-uniform
        unsigned long long pos0 = 0;
        if (lane == 0) pos0 = atomicAdd(&op->n_export, (unsigned long long)total);
        pos0 = __shfl(pos0, 0);
#pragma unroll
        for (int j = 0; j < kExportGroups; ++j) {
            if (!reserved_key(k[j])) {
                const uint64_t pos = pos0 + (uint64_t)__popcll(m[j] & ((1ull << lane) - 1));
                if (keys_out && pos < cap) keys_out[pos] = k[j];
            }
            uint64_t rest = m[j];
            uint64_t done = 0;
            while (rest) {  // wave-uniform
                uint64_t mm = rest;
                int p = -1;
                for (int q = 0; q <= tile; ++q) {
                    if (mm) { p = __ffsll((unsigned long long)mm) - 1; mm &= mm - 1; } else p = -1;
                }
                const uint64_t pos = pos0 + done + tile;
                if (p >= 0 && pos < cap) {
                    const uint64_t src = (c0 + (uint64_t)j * 64 + p) * dim4, dst = pos * dim4;
                    for (uint32_t c = tl; c < dim4; c += 16) {
                        if (values_out) values_out[dst + c] = values[src + c];
                        if (s1_out) s1_out[dst + c] = s1[src + c];
                        if (s2_out) s2_out[dst + c] = s2[src + c];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) rest &= rest - 1;  // x & (x-1) of 0 stays 0
                done += 4;
            }
            pos0 += (uint64_t)__popcll(m[j]);
        }
    }
}

}  // namespace mee

using namespace mee;

extern "C" {

int mee_reserve(mee_table* t, uint64_t new_capacity, void* stream) {
    MEE_RANGE("mee_reserve");
    if (!t || new_capacity == 0) return fail(MEE_ERR_INVALID_ARG, "mee_reserve: null table or zero capacity");
    if (t->prepared_n) {
        if (!t->prepared_by_forward) return fail(MEE_ERR_INVALID_ARG, "mee_reserve: a prepared apply is pending on this table (finish it with mee_apply_* or mee_apply_discard)");
        if (int rc = mee_apply_discard(t, stream)) return rc;   // a training forward's partition: the apply that follows partitions its batch again
    }
    size_t stored = 0;
    if (int rc = mee_size(t, &stored, stream)) return rc;
    const uint64_t nnb = next_prime((new_capacity + kW - 1) / kW), ncap = nnb * kW;
    if (ncap < stored) return fail(MEE_ERR_INVALID_ARG, "mee_reserve: capacity %llu is below the %zu stored keys", (unsigned long long)ncap, stored);
    if (nnb == t->nb) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    int64_t* nkeys = nullptr; uint32_t* nhits = nullptr;
    float *nv = nullptr, *n1 = nullptr, *n2 = nullptr;
    const uint64_t plane = ncap * (uint64_t)t->dim * sizeof(float);
    hipError_t e = hipMalloc((void**)&nkeys, ncap * sizeof(int64_t));
    if (e == hipSuccess && t->hits) e = hipMalloc((void**)&nhits, ncap * sizeof(uint32_t));
    if (e == hipSuccess) e = plane_alloc(t->value_memory, &nv, plane);
    if (e == hipSuccess && t->s1) e = plane_alloc(t->value_memory, &n1, plane);
    if (e == hipSuccess && t->s2) e = plane_alloc(t->value_memory, &n2, plane);
    if (e == hipSuccess) {
        fill_keys(nkeys, ncap, kEmpty, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess && nhits) e = hipMemsetAsync(nhits, 0, ncap * sizeof(uint32_t), st);
    if (e == hipSuccess) {
        rehash_kernel<<<grid_for(t->capacity, 256, 1u << 16), 256, 0, st>>>(t->keys, (const float4*)t->values, (const float4*)t->s1, (const float4*)t->s2,
                                                                          t->hits, t->capacity, nkeys, (float4*)nv, (float4*)n1, (float4*)n2, nhits, nnb,
                                                                          t->dim4, t->ctr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {   // the table is untouched
        (void)hipFree(nkeys); (void)hipFree(nhits);
        plane_free(t->value_memory, nv); plane_free(t->value_memory, n1); plane_free(t->value_memory, n2);
        return fail(e == hipErrorOutOfMemory ? MEE_ERR_OUT_OF_MEMORY : MEE_ERR_HIP, "mee_reserve(%llu slots): %s (old and new planes must fit together)",
                    (unsigned long long)ncap, hipGetErrorString(e));
    }
    (void)hipFree(t->keys); (void)hipFree(t->hits);
    plane_free(t->value_memory, t->values); plane_free(t->value_memory, t->s1); plane_free(t->value_memory, t->s2);
    t->keys = nkeys; t->hits = nhits; t->values = nv; t->s1 = n1; t->s2 = n2;
    t->nb = nnb; t->capacity = ncap;
    t->table_bytes = ncap * sizeof(int64_t) + (nhits ? ncap * sizeof(uint32_t) : 0) + plane * (1 + (n1 != nullptr) + (n2 != nullptr));
    ++t->generation;
    ++t->handle_epoch;
    return MEE_OK;
}

int mee_hits_scan(mee_table* t, uint32_t min_hits, uint32_t max_hits, int reset, int64_t* d_keys_out, size_t cap, size_t* n_out,
                  void* stream) {
    MEE_RANGE("mee_hits_scan");
    if (!t || !n_out || (cap && !d_keys_out)) return fail(MEE_ERR_INVALID_ARG, "mee_hits_scan: null argument");
    if (!t->hits) return fail(MEE_ERR_UNSUPPORTED, "mee_hits_scan: table was created without MEE_FLAG_TRACK_HITS");
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    zero_words(&t->op->n_export, sizeof(unsigned long long), st);
    hits_scan_kernel<<<grid_for(t->capacity, 4 * 1024, 4096), 256, 0, st>>>(t->keys, t->hits, t->capacity, min_hits, max_hits, reset, d_keys_out, cap, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)t->h_op->n_export;  // how many keys qualified; min(*n_out, cap) were written
    return MEE_OK;
}

int mee_export(const mee_table* t, int64_t* d_keys_out, float* d_values_out, float* d_state1_out, float* d_state2_out,
               size_t cap, size_t* n_out, void* stream) {
    MEE_RANGE("mee_export");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_export: null argument");
    return mee_export_range(t, 0, t->capacity, d_keys_out, d_values_out, d_state1_out, d_state2_out, cap, n_out, stream);
}

int mee_export_range(const mee_table* t, uint64_t slot_begin, uint64_t slot_end, int64_t* d_keys_out, float* d_values_out,
                     float* d_state1_out, float* d_state2_out, size_t cap, size_t* n_out, void* stream) {
    MEE_RANGE("mee_export_range");
    if (!t || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_export: null argument");
    if (slot_end > t->capacity) slot_end = t->capacity;
    if (slot_begin > slot_end) return fail(MEE_ERR_INVALID_ARG, "mee_export_range: slot_begin %llu > slot_end %llu", (unsigned long long)slot_begin, (unsigned long long)slot_end);
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    zero_words(&t->op->n_export, sizeof(unsigned long long), st);
    export_kernel<<<grid_for(slot_end - slot_begin, 4 * 64 * kExportGroups, 256 * 16), 256, 0, st>>>(
        t->keys, (const float4*)t->values, (const float4*)t->s1, (const float4*)t->s2, slot_begin, slot_end, t->dim4, d_keys_out,
        (float4*)d_values_out, t->s1 ? (float4*)d_state1_out : nullptr, t->s2 ? (float4*)d_state2_out : nullptr, cap, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)t->h_op->n_export;
    return MEE_OK;
}

static int read_counters(const mee_table* t, void* stream) {
    DeviceGuard g(t->device);
    MEE_HIP(hipMemcpyAsync(t->h_ctr, t->ctr, sizeof(Counters), hipMemcpyDeviceToHost, as_stream(stream)));
    MEE_HIP(hipStreamSynchronize(as_stream(stream)));
    return MEE_OK;
}
int mee_size(const mee_table* t, size_t* n_out, void* stream) {
    MEE_RANGE("mee_size");
    if (!t || !n_out) return fail(MEE_ERR_INVALID_ARG, "mee_size: null argument");
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    zero_words(&t->op->n_export, sizeof(unsigned long long), st);
    count_kernel<<<grid_for(t->capacity, 256 * 16, 2048), 256, 0, st>>>(t->keys, t->capacity, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *n_out = (size_t)t->h_op->n_export;
    return MEE_OK;
}
int mee_table_plane(const mee_table* t, uint32_t plane, void** ptr_out, uint64_t* row_stride_bytes, uint32_t* value_memory) {
    if (!t || !ptr_out) return fail(MEE_ERR_INVALID_ARG, "mee_table_plane: null argument");
    const float* p = plane_of(t, plane);
    if (!p) return fail(MEE_ERR_UNSUPPORTED, "mee_table_plane: plane %u does not exist (optimizer=%u)", plane, t->optimizer);
    *ptr_out = const_cast<float*>(p);
    if (row_stride_bytes) *row_stride_bytes = (uint64_t)t->dim * sizeof(float);
    if (value_memory) *value_memory = t->value_memory;
    return MEE_OK;
}

int mee_probe_length(const mee_table* t, const int64_t* d_keys, size_t n, uint64_t* buckets_visited_out, void* stream) {
    MEE_RANGE("mee_probe_length");
    if (!t || !buckets_visited_out || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_probe_length: null argument");
    *buckets_visited_out = 0;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    zero_words(&t->op->n_export, sizeof(unsigned long long), st);
    probe_length_kernel<<<grid_for(n, 16, 4096), 256, 0, st>>>(t->keys, t->nb, d_keys, n, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    *buckets_visited_out = (uint64_t)t->h_op->n_export;
    return MEE_OK;
}
int mee_probe_histogram(const mee_table* t, const int64_t* d_keys, size_t n, uint64_t* hist_out /* [4] */, void* stream) {
    MEE_RANGE("mee_probe_histogram");
    if (!t || !hist_out || (n && !d_keys)) return fail(MEE_ERR_INVALID_ARG, "mee_probe_histogram: null argument");
    for (int q = 0; q < 4; ++q) hist_out[q] = 0;
    if (n == 0) return MEE_OK;
    DeviceGuard g(t->device);
    hipStream_t st = as_stream(stream);
    zero_words(&t->op->n_export, sizeof(unsigned long long), st);
    zero_words(t->op->hist, sizeof t->op->hist, st);
    probe_length_kernel<<<grid_for(n, 16, 4096), 256, 0, st>>>(t->keys, t->nb, d_keys, n, t->op);
    MEE_HIP(hipGetLastError());
    MEE_HIP(hipMemcpyAsync(t->h_op, t->op, sizeof(OpCounters), hipMemcpyDeviceToHost, st));
    MEE_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < 4; ++q) hist_out[q] = (uint64_t)t->h_op->hist[q];
    return MEE_OK;
}
int mee_status(const mee_table* t, uint32_t* bits_out, void* stream) {
    MEE_RANGE("mee_status");
    if (!t || !bits_out) return fail(MEE_ERR_INVALID_ARG, "mee_status: null argument");
    if (int rc = read_counters(t, stream)) return rc;
    *bits_out = t->h_ctr->status;
    return MEE_OK;
}
int mee_clear_status(mee_table* t, void* stream) {
    MEE_RANGE("mee_clear_status");
    if (!t) return fail(MEE_ERR_INVALID_ARG, "mee_clear_status: null table");
    DeviceGuard g(t->device);
    zero_words(&t->ctr->status, sizeof(uint32_t), as_stream(stream));
    return MEE_OK;
}


}  // extern "C"

// meepo_host.h — host-side plumbing shared by the C-ABI translation units (error slot, device guard).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/meepo_embedding.h"

namespace mee {

char* last_error_buf();  // thread-local, 512 bytes (defined in meepo_table.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define MEE_HIP(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return ::mee::fail(e_ == hipErrorOutOfMemory ? MEE_ERR_OUT_OF_MEMORY : MEE_ERR_HIP,           \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// roctx ranges around every operator of the C-ABI (tracing), so that `rocprofv3 --marker-trace --kernel-trace` groups the kernels by the
// operator that launched them.  The marker library (librocprofiler-sdk-roctx.so.1, part of ROCm) is bound with dlopen at the first operator call; when
// it is not there — or MEE_ROCTX=0 — a range is one predictable branch.  Without a profiler attached a push / pop pair costs ~0.1 us of host time.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx& roctx_api();   // (meepo_table.hip)
struct OpRange {
    bool on;
    explicit OpRange(const char* name) { const Roctx& r = roctx_api(); on = r.push != nullptr; if (on) (void)r.push(name); }
    ~OpRange() { if (on) (void)roctx_api().pop(); }
    OpRange(const OpRange&) = delete;
    OpRange& operator=(const OpRange&) = delete;
};
#define MEE_RANGE(name) ::mee::OpRange mee_range_(name)

// Makes `device` current for the scope of one API call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// what other translation units may know about a table (defined in meepo_table.hip)
struct TableView {
    int device;
    const int64_t* keys;
    const float* values;
    uint64_t nb;
    uint32_t dim, dim4;
    float default_value;
    uint64_t generation;   // bumped whenever the planes move (mee_reserve)
    float *s1, *s2;        // optimizer planes (null without)
    uint32_t optimizer;
    // what creating a key needs (find_or_insert): initial row, initial state, the sticky status word, the hit counters
    uint32_t initializer;
    float init_scale, init_acc;
    uint64_t init_seed;
    uint32_t* status;
    uint32_t* hits;
};
TableView table_view(const mee_table* t);
int find_skip_padding(const mee_table* t, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream);   // meepo_find.hip

// ---- table groups (meepo_group.hip: grouped find / locate; meepo_table.hip: grouped apply) ---------------------------------
struct GroupDesc {   // 48 bytes per member table, device resident
    const int64_t* tkeys;
    float4* values;
    float4* s1;
    float4* s2;
    uint64_t nb;
    float defv;
    uint32_t pad;
};
struct GroupInit {   // per member, device resident: what creating a key in that table needs (grouped find_or_insert)
    uint64_t init_seed;
    uint32_t* status;
    uint32_t* hits;
    uint32_t initializer, optimizer;
    float init_scale, init_acc;
};
constexpr int kGroupSlotBits = 48;   // grouped apply names a row by (member << 48 | slot): unique per (table, key), never a reserved key
constexpr uint32_t kMaxGroupTables = 1024;   // offsets staged in LDS: (1024 + 1) x 8 B

inline unsigned grid_for(size_t work_items, unsigned per_block, unsigned cap) {
    size_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace mee

#include <vector>
struct mee_group {
    int device;
    uint32_t n_tables, dim, dim4, optimizer;
    std::vector<mee_table*> tables;
    std::vector<uint64_t> generations;   // of each table when its descriptor was last uploaded (mee_reserve moves planes)
    mee::GroupDesc* d_desc;
    mee::GroupInit* d_init;
    uint8_t* d_fmask;      // [max_apply_batch] found mask of grouped find_or_insert when the caller passes none
    // grouped apply (max_apply_batch > 0): a scratch-only table whose group table / per-position arrays serve the whole
    // jagged batch, and the located rows of the batch
    uint64_t max_apply_batch;
    mee_table* scratch;
    int64_t* d_gslot;
};
namespace mee {
int group_refresh(mee_group* g, void* stream);   // re-upload descriptors of members whose planes moved
int group_locate(mee_group* g, const int64_t* d_keys, const uint64_t* d_offsets, uint64_t off_stride, size_t n, int64_t* d_gslot, hipStream_t st);
}

// fake_rccl.cpp — TEST INFRASTRUCTURE, not product code.
//
// A stand-in for librccl.so.1 that carries ncclSend / ncclRecv messages between PROCESSES THAT SHARE ONE GPU through POSIX shared
// memory (device -> host mailbox -> device).  RCCL itself refuses several ranks on one device, and the GPU box of this pool has one GPU:
// with this library bound instead (MEE_RCCL_LIB=…/libfake_rccl.so, see csrc/meepo_sharded.hip) the exchange code behind the C-ABI —
// segment offsets, the counts exchange, exact and padded layouts, the grouped send/recv order, the way back — runs with 2-4 ranks and is
// checked against the oracle (tests/test_sharded.py).  It proves nothing about xGMI or RCCL; it proves the library's own bookkeeping
// for G > 1.  Only the twelve entry points the library binds are provided.
//
// Mailbox (src, dst): a ring of kSlots messages; the sender copies device memory into the next free slot and bumps `head`, the
// receiver waits for head > tail, copies out and bumps `tail`.  Inside ncclGroupStart/End all sends run first, then all receives:
// a rank never posts more than kSlots messages to one peer before that peer receives, so nobody waits on a full ring forever.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

// communicators this stand-in has freed (ncclCommAbort / ncclCommDestroy below); ranks may be threads of one process (sharded_mp_test … threads)
static std::mutex g_freed_mu;
static std::vector<void*> g_freed;
static void forget_freed(void* c) { std::lock_guard<std::mutex> lk(g_freed_mu); for (size_t i = 0; i < g_freed.size(); ++i) if (g_freed[i] == c) { g_freed.erase(g_freed.begin() + i); return; } }

namespace {

constexpr int kSlots = 4;
constexpr double kTimeoutS = 120.0;

struct Box {
    uint64_t head, tail;            // messages produced / consumed (acquire / release accesses: ld / st below)
    uint64_t bytes[kSlots];
    char pad[4096 - 16 - 8 * kSlots];
};
static_assert(sizeof(Box) == 4096, "Box is one page");

struct FakeComm {
    int rank, n, device;
    char name[64];
    char* base;
    size_t total, stride, slot_bytes;
    Box* box(int src, int dst) const { return reinterpret_cast<Box*>(base + ((size_t)src * n + dst) * stride); }
    char* data(Box* b, uint64_t seq) const { return reinterpret_cast<char*>(b) + sizeof(Box) + (size_t)(seq % kSlots) * slot_bytes; }
};

struct Op { bool send; void* buf; size_t bytes; int peer; FakeComm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

inline uint64_t ld(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void st(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
size_t slot_bytes_from_env() {
    const char* e = getenv("MEE_FAKE_RCCL_SLOT_MB");
    return (size_t)(e ? atoi(e) : 16) << 20;
}
size_t dt_size(ncclDataType_t dt) {
    switch (dt) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t do_send(const Op& o) {
    FakeComm* c = o.comm;
    if (o.bytes > c->slot_bytes) { fprintf(stderr, "fake_rccl: message of %zu bytes exceeds the slot (%zu): raise MEE_FAKE_RCCL_SLOT_MB\n", o.bytes, c->slot_bytes); return ncclInvalidArgument; }
    Box* b = c->box(c->rank, o.peer);
    const double t0 = now_s();
    while (ld(&b->head) - ld(&b->tail) >= kSlots) {
        if (now_s() - t0 > kTimeoutS) return ncclSystemError;
        usleep(50);
    }
    const uint64_t seq = ld(&b->head);
    if (hipMemcpy(c->data(b, seq), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    b->bytes[seq % kSlots] = o.bytes;
    st(&b->head, seq + 1);
    return ncclSuccess;
}
ncclResult_t do_recv(const Op& o) {
    FakeComm* c = o.comm;
    Box* b = c->box(o.peer, c->rank);
    const double t0 = now_s();
    while (ld(&b->head) == ld(&b->tail)) {
        if (now_s() - t0 > kTimeoutS) return ncclSystemError;
        usleep(50);
    }
    const uint64_t seq = ld(&b->tail);
    if (b->bytes[seq % kSlots] != o.bytes) {
        fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, the message holds %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)b->bytes[seq % kSlots]);
        return ncclInvalidUsage;
    }
    if (hipMemcpy(o.buf, c->data(b, seq), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    st(&b->tail, seq + 1);
    return ncclSuccess;
}
ncclResult_t run(std::vector<Op>& ops) {
    for (const Op& o : ops)   // what earlier kernels on the stream wrote must be in memory before it is copied out
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (const Op& o : ops) if (o.send) { ncclResult_t r = do_send(o); if (r != ncclSuccess) return r; }
    for (const Op& o : ops) if (!o.send) { ncclResult_t r = do_recv(o); if (r != ncclSuccess) return r; }
    return ncclSuccess;
}
ncclResult_t post(Op o) {
    if (!o.comm || o.peer < 0 || o.peer >= o.comm->n || o.peer == o.comm->rank) return ncclInvalidArgument;
    if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run(one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/meefake_%d_%ld", (int)getpid(), (long)(ts.tv_nsec ^ ts.tv_sec));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm* c = new FakeComm();
    c->rank = rank; c->n = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    strncpy(c->name, id.internal, sizeof c->name - 1);
    c->slot_bytes = slot_bytes_from_env();
    c->stride = sizeof(Box) + kSlots * c->slot_bytes;
    c->total = (size_t)nranks * nranks * c->stride;
    int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) {   // the creator sizes the segment; the kernel hands out zero pages: every ring starts empty
        if (ftruncate(fd, (off_t)c->total) != 0) { close(fd); delete c; return ncclSystemError; }
    } else {
        const double t0 = now_s();
        struct stat st;
        while (true) {
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == c->total) break;
            if (fd >= 0) close(fd);
            if (now_s() - t0 > kTimeoutS) { delete c; return ncclSystemError; }
            usleep(1000);
        }
    }
    c->base = (char*)mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    forget_freed(c);   // (the allocator may hand out an address a destroyed communicator had)
    return ncclSuccess;
}

// communicators this stand-in has freed: destroying or aborting one of them again is the double free the real library would commit —
// reported loudly (the test that drives the abort path fails on it)
// (g_freed / forget_freed: defined in front of the anonymous namespace's users, above)
static void note_free(void* c, const char* who) {
    std::lock_guard<std::mutex> lk(g_freed_mu);
    for (void* f : g_freed)
        if (f == c) { fprintf(stderr, "fake_rccl: %s on a communicator that was already freed (double free)\n", who); abort(); }
    g_freed.push_back(c);
}

// ncclCommAbort frees the communicator, like ncclCommDestroy
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c) return ncclSuccess;
    note_free(c, "ncclCommAbort");
    munmap(c->base, c->total);
    shm_unlink(c->name);
    c->base = nullptr;   // (the object itself stays allocated: a use after the abort is caught below instead of corrupting the heap)
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c) return ncclSuccess;
    note_free(c, "ncclCommDestroy");
    munmap(c->base, c->total);
    shm_unlink(c->name);   // the first rank to get here removes the name; the others keep their mapping until they unmap
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = reinterpret_cast<const FakeComm*>(comm)->n; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { *rank = reinterpret_cast<const FakeComm*>(comm)->rank; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device) { *device = reinterpret_cast<const FakeComm*>(comm)->device; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    // test hook: the MEE_FAKE_RCCL_FAIL_GROUP-th outermost ncclGroupEnd of the process (1-based) fails — the library's abort path
    static std::atomic<int> n_groups{0};   // (ranks may be threads of one process)
    static const int fail_at = getenv("MEE_FAKE_RCCL_FAIL_GROUP") ? atoi(getenv("MEE_FAKE_RCCL_FAIL_GROUP")) : 0;
    if (++n_groups == fail_at) { g_ops.clear(); return ncclSystemError; }
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!reinterpret_cast<FakeComm*>(comm)->base) { fprintf(stderr, "fake_rccl: ncclSend on an aborted communicator (use after free)\n"); abort(); }
    return post(Op{true, const_cast<void*>(sendbuff), count * dt_size(datatype), peer, reinterpret_cast<FakeComm*>(comm), stream});
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!reinterpret_cast<FakeComm*>(comm)->base) { fprintf(stderr, "fake_rccl: ncclRecv on an aborted communicator (use after free)\n"); abort(); }
    return post(Op{false, recvbuff, count * dt_size(datatype), peer, reinterpret_cast<FakeComm*>(comm), stream});
}

// uint64 sums only (mee_sharded_size): every rank sends its values to every other rank and adds up what it receives
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (datatype != ncclUint64 || op != ncclSum || count > 64) return ncclInvalidArgument;
    // test hook: the MEE_FAKE_RCCL_FAIL_ALLREDUCE-th ncclAllReduce of the process (1-based) fails — the library's abort path at world 1
    static std::atomic<int> n_allreduce{0};
    static const int fail_ar = getenv("MEE_FAKE_RCCL_FAIL_ALLREDUCE") ? atoi(getenv("MEE_FAKE_RCCL_FAIL_ALLREDUCE")) : 0;
    if (++n_allreduce == fail_ar) return ncclSystemError;
    if (!c->base) { fprintf(stderr, "fake_rccl: ncclAllReduce on an aborted communicator (use after free)\n"); abort(); }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    uint64_t mine[64], sum[64];
    if (hipMemcpy(mine, sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    memcpy(sum, mine, count * 8);
    void* tmp = nullptr;
    if (hipMalloc(&tmp, count * 8) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < c->n && r == ncclSuccess; ++p)
        if (p != c->rank) r = do_send(Op{true, const_cast<void*>(sendbuff), count * 8, p, c, stream});
    for (int p = 0; p < c->n && r == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        r = do_recv(Op{false, tmp, count * 8, p, c, stream});
        uint64_t got[64];
        if (r == ncclSuccess && hipMemcpy(got, tmp, count * 8, hipMemcpyDeviceToHost) != hipSuccess) r = ncclUnhandledCudaError;
        for (size_t i = 0; i < count && r == ncclSuccess; ++i) sum[i] += got[i];
    }
    (void)hipFree(tmp);
    if (r == ncclSuccess && hipMemcpy(recvbuff, sum, count * 8, hipMemcpyHostToDevice) != hipSuccess) r = ncclUnhandledCudaError;
    return r;
}

const char* ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: shared-memory set-up failed or a peer did not answer within the time-out";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        case ncclInvalidUsage: return "fake_rccl: invalid usage (message size mismatch / unbalanced group)";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"

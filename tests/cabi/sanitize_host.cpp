// sanitize_host.cpp — TEST INFRASTRUCTURE: drives the HOST side of the C-ABI under AddressSanitizer / ThreadSanitizer in the CPU container.
//
// Linked against the host-only build of csrc/*.hip, tests/cabi/hip_host_stub.cpp (device memory = host memory, launches do nothing) and — through
// MEE_RCCL_LIB — tests/cabi/fake_rccl.cpp.  G rank THREADS (default 8: the node's GPU count) each create a communicator handle, tables (a hot one, a cold one,
// one per context flavour) and sharded contexts (exact, padded, padded + dedup, exact + dedup over a hot/cold pair), run every mee_sharded_* operator a few
// times, the single-table operators the contexts are built on, routers and peer contexts, and tear everything down.  Kernels do not run, so no VALUE is
// checked — only return codes: what the sanitizers watch is the library's own bookkeeping (offsets and sizes of every copy, lifetimes, the process-wide state
// rank threads share: the lazy RCCL / roctx binders, the abort registry, the calibration cache, the per-table output ring, the thread-local error slot).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "meepo_embedding.h"

#define MEECK(x) do { int rc_ = (x); if (rc_ != MEE_OK) { fprintf(stderr, "[rank %d] %s -> %d: %s (line %d)\n", rank, #x, rc_, mee_last_error(), __LINE__); return 3; } } while (0)

static char g_id[MEE_COMM_ID_BYTES];
static std::atomic<int> g_id_ready{0};

static mee_table* make_table(uint64_t cap, uint32_t dim, uint64_t max_batch, uint32_t mem) {
    mee_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c; c.capacity = cap; c.dim = dim; c.optimizer = MEE_OPT_ADAGRAD; c.max_batch = max_batch; c.value_memory = mem; c.flags = MEE_FLAG_TRACK_HITS;
    mee_table* t = nullptr;
    return mee_table_create(&c, &t) == MEE_OK ? t : nullptr;
}

static int run_rank(int rank, int G) {
    if (rank == 0) {
        MEECK(mee_comm_unique_id(g_id));
        g_id_ready.store(1, std::memory_order_release);
    } else while (!g_id_ready.load(std::memory_order_acquire)) std::this_thread::yield();
    void* comm = nullptr;
    MEECK(mee_comm_create(g_id, (uint32_t)G, (uint32_t)rank, 0, &comm));
    const uint32_t dim = 64;
    const size_t B = 4096;
    std::vector<int64_t> keys(B);
    for (size_t i = 0; i < B; ++i) keys[i] = (int64_t)(0x9E3779B97F4A7C15ull * (i + 1 + (size_t)rank * B));
    std::vector<float> rows(B * dim, 0.25f), out(B * dim);
    std::vector<uint8_t> found(B);
    std::vector<int64_t> slots(B), uniq(B), inverse(B);
    std::vector<uint32_t> counts(B);
    struct Flavour { double slack; uint32_t flags; bool cold; } flavours[] = {{0.0, 0u, false}, {1.5, 0u, false}, {1.25, MEE_SHARDED_DEDUP, false}, {0.0, MEE_SHARDED_DEDUP, true}};
    for (const Flavour& f : flavours) {
        mee_table* hot = make_table(1 << 15, dim, (uint64_t)G * (B + 2048), MEE_MEM_HBM);
        mee_table* cold = f.cold ? make_table(1 << 15, dim, (uint64_t)G * (B + 2048), MEE_MEM_HOST_PINNED) : nullptr;
        if (!hot || (f.cold && !cold)) { fprintf(stderr, "[rank %d] table: %s\n", rank, mee_last_error()); return 4; }
        mee_sharded_options o;
        memset(&o, 0, sizeof o);
        o.struct_size = sizeof o; o.max_batch = B; o.pad_slack = f.slack; o.flags = f.flags; o.cold = cold; o.hot_key_limit = 1000;
        mee_sharded* s = nullptr;
        MEECK(mee_sharded_create_ex(hot, comm, &o, &s));
        for (int rep = 0; rep < 3; ++rep) {
            const size_t n = rep == 1 ? 0 : B - (size_t)rep * 17;   // (an empty batch on a rank is still a collective call)
            MEECK(mee_sharded_insert(s, keys.data(), rows.data(), n, nullptr));
            MEECK(mee_sharded_find(s, keys.data(), n, out.data(), found.data(), nullptr));
            MEECK(mee_sharded_find_or_insert(s, keys.data(), n, out.data(), found.data(), nullptr));
            MEECK(mee_sharded_assign(s, keys.data(), rows.data(), n, found.data(), nullptr));
            MEECK(mee_sharded_apply_adagrad(s, keys.data(), rows.data(), n, 0.01f, 1e-10f, nullptr));
            MEECK(mee_sharded_remove(s, keys.data(), n / 2, found.data(), nullptr));
            size_t total = 0;
            MEECK(mee_sharded_size(s, &total, nullptr));
            uint32_t bits = 0;
            MEECK(mee_sharded_status(s, &bits, nullptr));
            MEECK(mee_sharded_clear_status(s, nullptr));
        }
        // the single-table operators the owner side is made of, on this rank's own tables (another thread's tables are never touched)
        MEECK(mee_find(hot, keys.data(), B, out.data(), found.data(), nullptr));
        MEECK(mee_find_ex(hot, keys.data(), B, out.data() + (rank & 1) * dim, found.data(), MEE_FIND_STREAM_ROWS, nullptr));
        MEECK(mee_find_located_prepare(hot, keys.data(), B, out.data(), found.data(), slots.data(), nullptr));
        MEECK(mee_apply_adagrad_located(hot, keys.data(), slots.data(), rows.data(), B, 0.01f, 1e-10f, nullptr));
        MEECK(mee_dedup_keys(hot, keys.data(), B, uniq.data(), inverse.data(), -1, nullptr));
        MEECK(mee_dedup_sum(hot, keys.data(), rows.data(), B, uniq.data(), out.data(), counts.data(), inverse.data(), -1, nullptr));
        MEECK(mee_assign(hot, keys.data(), rows.data(), B, found.data(), nullptr));
        MEECK(mee_reserve(hot, 1 << 16, nullptr));
        uint64_t hist[4];
        MEECK(mee_probe_histogram(hot, keys.data(), B, hist, nullptr));
        MEECK(mee_sharded_destroy(s));
        MEECK(mee_table_destroy(hot));
        if (cold) MEECK(mee_table_destroy(cold));
    }
    // routers and the peer-mapped contexts' bookkeeping (handles of ALL ranks are needed to connect: only the local half is exercised here)
    mee_router* r = nullptr;
    MEECK(mee_router_create(0, B, (uint32_t)G, &r));
    std::vector<uint64_t> cnt((size_t)G);
    MEECK(mee_partition(r, keys.data(), B, uniq.data(), cnt.data(), inverse.data(), nullptr));
    MEECK(mee_partition_padded(r, keys.data(), B, uniq.data(), cnt.data(), inverse.data(), nullptr));
    mee_p2p* p = nullptr;
    MEECK(mee_p2p_create(0, (uint32_t)G, (uint32_t)rank, 1024, B, dim, 1, &p));
    char handles[MEE_P2P_BUFFERS * MEE_IPC_HANDLE_BYTES];
    MEECK(mee_p2p_export(p, handles));
    MEECK(mee_p2p_destroy(p));
    MEECK(mee_router_destroy(r));
    mee_calibration cal;
    cal.struct_size = sizeof cal;
    MEECK(mee_device_calibration(0, &cal));
    if (mee_find(nullptr, nullptr, 1, nullptr, nullptr, nullptr) != MEE_ERR_INVALID_ARG || !strstr(mee_last_error(), "mee_find")) return 5;   // the error slot is this thread's
    MEECK(mee_comm_destroy(comm));
    return 0;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 8;
    if (G < 1 || G > 16) return 64;
    if (!getenv("MEE_RCCL_LIB")) { fprintf(stderr, "sanitize_host: MEE_RCCL_LIB must name the stand-in for librccl\n"); return 65; }
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::thread> ts;
    for (int r = 0; r < G; ++r) ts.emplace_back([&, r] { rcs[(size_t)r] = run_rank(r, G); });
    for (auto& t : ts) t.join();
    int worst = 0;
    for (int rc : rcs) if (rc > worst) worst = rc;
    if (worst == 0) printf("sanitize_host ok: %d rank threads through the host side of the C-ABI\n", G);
    return worst;
}

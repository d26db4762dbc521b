// sharded_mp_test.cpp — the row-sharded table through the C-ABI with G > 1 ranks, from plain C++ (no Python, no torch).
//
// usage: sharded_mp_test [G] [threads]        (default 2 ranks, one process each)
// `threads`: the G ranks are THREADS of this one process, every one with a context, a table and a communicator handle of its own on device 0 (the stand-in
// for librccl keeps its group state per thread) — how G = 8, the node's GPU count, runs on a box that lets one job keep at most 6 processes on its card.
// The parent forks G rank processes BEFORE any HIP call (a forked child of a process that has initialised HIP is unusable); rank r uses
// device r when the box has >= G GPUs (then the library binds the real RCCL: ranks exchange over xGMI), else every rank uses device 0 and
// the environment must name a stand-in for librccl in MEE_RCCL_LIB (RCCL itself refuses several ranks on one device; the test suite's
// tests/cabi/fake_rccl.cpp carries ncclSend/ncclRecv through shared memory).  Rank 0 makes the ncclUniqueId and hands it to the others
// through a shared page.
//
// Sequence per rank (every mee_sharded_* call is collective): insert its slice of N keys with key-derived rows -> size == N -> find a
// permutation of ALL keys (+ absent ones) -> one sparse-Adagrad step on its slice -> find again and compare with the update computed on the
// host -> a SKEWED step (a quarter of the slice twice more, with other gradients: with MEE_SHARDED_DEDUP the rank sends one summed row per distinct key) ->
// remove half of its slice -> size, found masks.  Twice: exact segments, and padded segments with pre-exchange dedup (mee_sharded_create_ex).
// Exit code 0 = every rank passed.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "meepo_embedding.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MEECK(x) do { int rc_ = (x); if (rc_ != MEE_OK) { fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, mee_last_error()); return 3; } } while (0)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "[rank %d] CHECK failed: %s (line %d)\n", g_rank, #c, __LINE__); return 4; } } while (0)

static thread_local int g_rank = -1;
static uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
static float row_value(int64_t key, int j, uint64_t seed) { return (float)(mix64((uint64_t)key ^ mix64(seed + j)) >> 40) * 0x1p-24f - 0.5f; }

struct Shared {            // one page shared by the parent and the ranks
    volatile int id_ready;
    char id[MEE_COMM_ID_BYTES];
};

template <typename T> struct Dev {
    T* p = nullptr;
    explicit Dev(size_t n) { if (hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)) != hipSuccess) p = nullptr; }
    ~Dev() { if (p) (void)hipFree(p); }
    void up(const std::vector<T>& h) { (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); }
    std::vector<T> down(size_t m) const { std::vector<T> h(m); (void)hipMemcpy(h.data(), p, m * sizeof(T), hipMemcpyDeviceToHost); return h; }
};

static int run_rank(int rank, int G, Shared* sh) {
    g_rank = rank;
    int ndev = 0;
    HIPCK(hipGetDeviceCount(&ndev));
    CHECK(ndev >= 1);
    const int device = ndev >= G ? rank : 0;
    if (ndev < G && !getenv("MEE_RCCL_LIB")) {
        fprintf(stderr, "[rank %d] %d ranks on %d GPU(s) need MEE_RCCL_LIB (a stand-in for librccl: RCCL refuses several ranks per device)\n", rank, G, ndev);
        return 5;
    }
    HIPCK(hipSetDevice(device));
    if (rank == 0) {
        MEECK(mee_comm_unique_id(sh->id));
        __sync_synchronize();
        sh->id_ready = 1;
    } else {
        for (int spins = 0; !sh->id_ready; ++spins) { if (spins > 600000) { fprintf(stderr, "[rank %d] no communicator id after 60 s\n", rank); return 6; } usleep(100); }
        __sync_synchronize();
    }
    void* comm = nullptr;
    MEECK(mee_comm_create(sh->id, (uint32_t)G, (uint32_t)rank, device, &comm));

    const uint32_t dim = 64;
    const size_t N = 60000, per = N / G, n_abs = 500;
    const float lr = 0.05f, eps = 1e-10f, acc0 = 0.1f;
    std::vector<int64_t> all(N + n_abs);
    for (size_t i = 0; i < N + n_abs; ++i) all[i] = (int64_t)mix64(11 + (i + 1) * 0x9E3779B97F4A7C15ull);   // the last n_abs are never inserted
    const size_t lo = (size_t)rank * per, cnt = rank == G - 1 ? N - lo : per;

    for (int variant = 0; variant < 2; ++variant) {
        mee_config c{};
        c.struct_size = sizeof c; c.device = device; c.capacity = (uint64_t)(N / G * 2.0); c.dim = dim; c.optimizer = MEE_OPT_ADAGRAD;
        c.max_batch = 2 * (N + n_abs); c.default_value = -3.0f; c.initial_accumulator = acc0;
        mee_table* table = nullptr;
        MEECK(mee_table_create(&c, &table));
        mee_sharded_options o{};
        o.struct_size = sizeof o; o.max_batch = N + n_abs; o.pad_slack = variant ? 1.5 : 0.0; o.flags = variant ? MEE_SHARDED_DEDUP : 0u;
        mee_sharded* s = nullptr;
        MEECK(mee_sharded_create_ex(table, comm, &o, &s));
        uint32_t gg = 0, rr = 99; uint64_t seg = 0;
        MEECK(mee_sharded_info(s, &gg, &rr, &seg));
        CHECK(gg == (uint32_t)G && rr == (uint32_t)rank && (seg != 0) == (variant == 1));

        // ---- insert my slice ----
        std::vector<int64_t> mine(all.begin() + lo, all.begin() + lo + cnt);
        std::vector<float> rows(cnt * dim), grads(cnt * dim);
        for (size_t i = 0; i < cnt; ++i)
            for (uint32_t j = 0; j < dim; ++j) { rows[i * dim + j] = row_value(mine[i], (int)j, 2); grads[i * dim + j] = 0.02f * row_value(mine[i], (int)j, 6); }
        Dev<int64_t> d_mine(cnt); Dev<float> d_rows(cnt * dim), d_grads(cnt * dim);
        CHECK(d_mine.p && d_rows.p && d_grads.p);
        d_mine.up(mine); d_rows.up(rows); d_grads.up(grads);
        MEECK(mee_sharded_insert(s, d_mine.p, d_rows.p, cnt, nullptr));
        size_t total = 0;
        MEECK(mee_sharded_size(s, &total, nullptr));
        CHECK(total == N);
        size_t local = 0;
        MEECK(mee_size(table, &local, nullptr));
        CHECK(local > N / G / 2 && local < N / G * 2);   // the keys spread over the shards

        // ---- find a permutation of ALL keys, the absent ones in between, and a few duplicates ----
        const size_t nq = N + n_abs;
        std::vector<int64_t> q(nq);
        for (size_t i = 0; i < nq; ++i) q[i] = all[(i * 7919 + (size_t)rank * 13) % nq];
        for (size_t i = 0; i < 200; ++i) q[i * 3 + 1] = q[7];   // duplicates of one key
        Dev<int64_t> d_q(nq); Dev<float> d_out(nq * dim); Dev<uint8_t> d_f(nq);
        CHECK(d_q.p && d_out.p && d_f.p);
        d_q.up(q);
        MEECK(mee_sharded_find(s, d_q.p, nq, d_out.p, d_f.p, nullptr));
        HIPCK(hipDeviceSynchronize());
        auto out = d_out.down(nq * dim); auto f = d_f.down(nq);
        // (which keys are stored: index in `all` below N)
        for (size_t i = 0; i < nq; ++i) {
            size_t idx = (i * 7919 + (size_t)rank * 13) % nq;
            if (i % 3 == 1 && i / 3 < 200) idx = (7 * 7919 + (size_t)rank * 13) % nq;
            const bool stored = idx < N;
            CHECK(f[i] == (stored ? 1 : 0));
            for (uint32_t j = 0; j < dim; j += 7) CHECK(out[i * dim + j] == (stored ? row_value(all[idx], (int)j, 2) : -3.0f));
        }

        // ---- one sparse-Adagrad step on my slice; every key is updated exactly once (the slices are disjoint) ----
        MEECK(mee_sharded_apply_adagrad(s, d_mine.p, d_grads.p, cnt, lr, eps, nullptr));
        MEECK(mee_sharded_size(s, &total, nullptr));   // a collective after the apply: every rank's pairs have been applied before anybody looks
        MEECK(mee_sharded_find(s, d_mine.p, cnt, d_out.p, d_f.p, nullptr));
        HIPCK(hipDeviceSynchronize());
        out = d_out.down(cnt * dim); f = d_f.down(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            CHECK(f[i] == 1);
            for (uint32_t j = 0; j < dim; j += 5) {
                const float g = grads[i * dim + j], an = fmaf(g, g, acc0), qv = g / (sqrtf(an) + eps), w = fmaf(-lr, qv, rows[i * dim + j]);
                CHECK(fabsf(out[i * dim + j] - w) <= 1e-6f * fabsf(w) + 1e-9f);
            }
        }

        // ---- a skewed step: the first quarter of my slice THREE times in one batch (gradients g, 2g and -g/2: their sum is 2.5 g), the rest once.  The
        // dedup context adds a key's three rows up on this rank before they travel; the owner applies one update per key either way ----
        {
            const size_t quarter = cnt / 4, n2 = cnt + 2 * quarter;
            std::vector<int64_t> k2(mine);
            std::vector<float> g2(grads);
            k2.insert(k2.end(), mine.begin(), mine.begin() + quarter); k2.insert(k2.end(), mine.begin(), mine.begin() + quarter);
            g2.resize(n2 * dim);
            for (size_t i = 0; i < quarter; ++i)
                for (uint32_t j = 0; j < dim; ++j) { g2[(cnt + i) * dim + j] = 2.0f * grads[i * dim + j]; g2[(cnt + quarter + i) * dim + j] = -0.5f * grads[i * dim + j]; }
            Dev<int64_t> d_k2(n2); Dev<float> d_g2(n2 * dim);
            CHECK(d_k2.p && d_g2.p);
            d_k2.up(k2); d_g2.up(g2);
            MEECK(mee_sharded_apply_adagrad(s, d_k2.p, d_g2.p, n2, lr, eps, nullptr));
            MEECK(mee_sharded_size(s, &total, nullptr));
            MEECK(mee_sharded_find(s, d_mine.p, cnt, d_out.p, d_f.p, nullptr));
            HIPCK(hipDeviceSynchronize());
            out = d_out.down(cnt * dim);
            for (size_t i = 0; i < cnt; ++i)
                for (uint32_t j = 0; j < dim; j += 5) {
                    const float g = grads[i * dim + j], a1 = fmaf(g, g, acc0), w1 = fmaf(-lr, g / (sqrtf(a1) + eps), rows[i * dim + j]);   // after the first step
                    const float gs = i < quarter ? (float)((double)g + (double)(2.0f * g) + (double)(-0.5f * g)) : g;
                    const float a2 = fmaf(gs, gs, a1), w2 = fmaf(-lr, gs / (sqrtf(a2) + eps), w1);
                    CHECK(fabsf(out[i * dim + j] - w2) <= 2e-6f * fabsf(w2) + 1e-9f);
                }
        }

        // ---- remove the first half of my slice ----
        const size_t half = cnt / 2;
        MEECK(mee_sharded_remove(s, d_mine.p, half, d_f.p, nullptr));
        HIPCK(hipDeviceSynchronize());
        f = d_f.down(half);
        for (size_t i = 0; i < half; ++i) CHECK(f[i] == 1);
        MEECK(mee_sharded_size(s, &total, nullptr));
        size_t removed = 0;
        for (int r2 = 0; r2 < G; ++r2) removed += ((r2 == G - 1 ? N - (size_t)r2 * per : per)) / 2;
        CHECK(total == N - removed);
        MEECK(mee_sharded_find(s, d_mine.p, cnt, d_out.p, d_f.p, nullptr));
        HIPCK(hipDeviceSynchronize());
        f = d_f.down(cnt);
        for (size_t i = 0; i < cnt; ++i) CHECK(f[i] == (i < half ? 0 : 1));
        uint32_t bits = 7;
        MEECK(mee_sharded_status(s, &bits, nullptr));
        CHECK(bits == 0);
        MEECK(mee_status(table, &bits, nullptr));
        CHECK(bits == 0);
        MEECK(mee_sharded_destroy(s));
        MEECK(mee_table_destroy(table));
    }
    MEECK(mee_comm_destroy(comm));
    return 0;
}

// a crash of this program must say where: the test runner only sees the exit status otherwise
static void on_crash(int sig) {
    void* frames[48];
    const int n = backtrace(frames, 48);
    const char msg[] = "sharded_mp_test: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char** argv) {
    signal(SIGSEGV, on_crash); signal(SIGBUS, on_crash); signal(SIGABRT, on_crash); signal(SIGFPE, on_crash);
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int G = argc > 1 ? atoi(argv[1]) : 2;
    const bool threads = argc > 2 && !strcmp(argv[2], "threads");
    if (G < 1 || G > 8) { fprintf(stderr, "usage: sharded_mp_test [G = 1..8] [threads]\n"); return 64; }
    Shared* sh = (Shared*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { perror("mmap"); return 65; }
    memset((void*)sh, 0, sizeof *sh);
    if (threads) {   // one process, G rank threads
        std::vector<int> rcs(G, 0);
        std::vector<std::thread> ts;
        for (int r = 0; r < G; ++r) ts.emplace_back([&, r] { rcs[r] = run_rank(r, G, sh); });
        for (auto& t : ts) t.join();
        int worst = 0;
        for (int rc : rcs) if (rc > worst) worst = rc;
        if (worst == 0) printf("sharded_mp_test ok: %d rank THREADS through mee_sharded_* (exact segments; padded segments + pre-exchange dedup)\n", G);
        return worst;
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < G; ++r) {   // fork first, HIP later: nothing in this process has touched the GPU
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 66; }
        if (pid == 0) _exit(run_rank(r, G, sh));
        kids.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : kids) {
        int st = 0;
        if (waitpid(pid, &st, 0) < 0) { worst = 67; continue; }
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (rc > worst) worst = rc;
    }
    if (worst == 0) printf("sharded_mp_test ok: %d ranks through mee_sharded_* (exact segments; padded segments + pre-exchange dedup)\n", G);
    return worst;
}

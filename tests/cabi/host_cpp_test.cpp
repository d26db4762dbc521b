// host_cpp_test.cpp — the C++ host layer (include/meepo_embedding.hpp) end to end on one GPU, no Python:
// Table verbs, a hot/cold TieredTable (rows of the cold half in pinned host DRAM), and the sharded pipeline
// (Router partition -> PeerExchange push -> find -> rows in batch order) with one rank.  Exit code 0 = pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "meepo_embedding.hpp"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 4; } } while (0)

static uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
static float row_value(int64_t key, int j) { return (float)(mix64((uint64_t)key ^ mix64(2 + j)) >> 40) * 0x1p-24f - 0.5f; }

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    explicit DevBuf(size_t n_) : n(n_) { if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) p = nullptr; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    void up(const std::vector<T>& h) { (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); }
    std::vector<T> down(size_t m) const { std::vector<T> h(m); (void)hipMemcpy(h.data(), p, m * sizeof(T), hipMemcpyDeviceToHost); return h; }
};

int main() try {
    const uint32_t dim = 64;
    const size_t n = 60000;
    std::vector<int64_t> keys(n);
    for (size_t i = 0; i < n; ++i) keys[i] = (int64_t)mix64(7 + (i + 1) * 0x9E3779B97F4A7C15ull);
    std::vector<float> rows(n * dim);
    for (size_t i = 0; i < n; ++i) for (uint32_t j = 0; j < dim; ++j) rows[i * dim + j] = row_value(keys[i], j);
    DevBuf<int64_t> d_keys(n); DevBuf<float> d_rows(n * dim), d_out(n * dim); DevBuf<uint8_t> d_found(n);
    CHECK(d_keys.p && d_rows.p && d_out.p && d_found.p);
    d_keys.up(keys); d_rows.up(rows);

    // ---- bad options surface as exceptions with the library's message -------------------------------------------
    bool threw = false;
    try { meepo::TableOptions bad; bad.capacity = 0; meepo::Table t(bad); } catch (const meepo::Error& e) { threw = e.code() == MEE_ERR_INVALID_ARG; }
    CHECK(threw);

    // ---- one table: insert, find, remove, size -------------------------------------------------------------------
    meepo::TableOptions o; o.capacity = (uint64_t)(n / 0.75); o.dim = dim; o.max_batch = n; o.default_value = -2.0f;
    meepo::Table table(o);
    table.insert(d_keys.p, d_rows.p, n);
    CHECK(table.size() == n && table.status() == 0);
    table.find(d_keys.p, n, d_out.p, d_found.p);
    HIPCK(hipDeviceSynchronize());
    { auto out = d_out.down(n * dim); auto f = d_found.down(n);
      for (size_t i = 0; i < n; ++i) CHECK(f[i] == 1);
      CHECK(memcmp(out.data(), rows.data(), n * dim * 4) == 0); }
    table.remove(d_keys.p, n / 2, d_found.p);
    CHECK(table.size() == n - n / 2);
    { // ranged export: the two halves of the slot range hold all pairs between them
      const uint64_t cap = table.info().capacity;
      const size_t a = table.export_range(0, cap / 2, nullptr, nullptr, 0), b = table.export_range(cap / 2, cap, nullptr, nullptr, 0);
      CHECK(a + b == n - n / 2 && a > 0 && b > 0);
      table.reserve(cap * 2);                                        // in-place rehash: capacity doubles, content stays
      CHECK(table.info().capacity >= cap * 2 && table.size() == n - n / 2 && table.status() == 0); }

    // ---- group: two tables, one launch ------------------------------------------------------------------------------
    { meepo::Table other(o);
      other.insert(d_keys.p, d_rows.p, n / 4);                       // holds only the first quarter of the keys
      meepo::Table* members[2] = {&other, &table};
      meepo::Group grp(members, 2);
      std::vector<uint64_t> offs = {0, n / 2, n};                    // first half of the batch -> `other`, second half -> `table`
      DevBuf<uint64_t> d_offs(3); d_offs.up(offs);
      grp.find(d_keys.p, d_offs.p, n, d_out.p, d_found.p);
      HIPCK(hipDeviceSynchronize());
      auto f = d_found.down(n); auto out = d_out.down(n * dim);
      // `other` holds keys [0, n/4); `table` lost keys [0, n/2) to the remove above and still holds [n/2, n)
      for (size_t i = 0; i < n; ++i) {
          const bool expect = i < n / 4 || i >= n / 2;
          CHECK(f[i] == (expect ? 1 : 0));
          if (expect) CHECK(memcmp(&out[i * dim], &rows[i * dim], dim * 4) == 0); else CHECK(out[i * dim] == -2.0f);
      } }

    // ---- one training step from C++: the located forward (its launch also partitions the batch for the backward), then sparse Adagrad at
    //      the slots it found — every key twice in the batch, grads = 1: w -= lr * 2 / sqrt(acc0 + 4) ---------------------------------
    { const size_t m = 20000;
      meepo::TableOptions to = o; to.optimizer = MEE_OPT_ADAGRAD; to.initial_accumulator = 0.5f; to.max_batch = 2 * m; to.capacity = 65536;
      meepo::Table tr(to);
      tr.insert(d_keys.p, d_rows.p, m);
      std::vector<int64_t> b(2 * m);
      for (size_t i = 0; i < m; ++i) { b[i] = keys[i]; b[m + i] = keys[(i * 31) % m]; }   // 31 is coprime to m: every key exactly twice
      b[7] = (int64_t)mix64(424242);                                  // an absent key: handle -1, no update (and keys[7] occurs once)
      DevBuf<int64_t> d_b(2 * m), d_slots(2 * m); DevBuf<float> d_o(2 * m * dim), d_g(2 * m * dim); DevBuf<uint8_t> d_f(2 * m);
      d_b.up(b); d_g.up(std::vector<float>(2 * m * dim, 1.0f));
      tr.find_located(d_b.p, 2 * m, d_o.p, d_f.p, d_slots.p, /*prepare=*/true);
      // a mutator between the training forward and its backward drops the partition the forward left; the apply partitions the batch again
      // (same result: the four rows are rewritten with what they hold)
      tr.insert(d_keys.p, d_rows.p, 4);
      tr.apply_adagrad_located(d_b.p, d_slots.p, d_g.p, 2 * m, 0.1f);
      CHECK(tr.status() == 0);
      tr.find(d_keys.p, m, d_o.p, d_f.p);
      HIPCK(hipDeviceSynchronize());
      auto out = d_o.down(m * dim); auto sl = d_slots.down(2 * m);
      CHECK(sl[7] == -1 && sl[8] >= 0);
      for (size_t i = 0; i < m; ++i) {
          const float gsum = i == 7 ? 1.0f : 2.0f, expect = rows[i * dim + 3] - 0.1f * gsum / sqrtf(0.5f + gsum * gsum);
          CHECK(fabsf(out[i * dim + 3] - expect) <= 1e-6f * fabsf(expect) + 1e-7f);
      }
      tr.find_located(d_b.p, 2 * m, d_o.p, d_f.p, d_slots.p, true);   // a partition nobody uses ...
      tr.apply_discard();                                              // ... is dropped
      tr.insert(d_keys.p, d_rows.p, 4);
      CHECK(tr.status() == 0); }

    // ---- hot/cold pair: first half of the keys in HBM, second half in the pinned-host tier ------------------------
    meepo::TableOptions ho = o; ho.capacity = (uint64_t)(n / 2 / 0.75);
    meepo::TableOptions co = ho; co.value_memory = MEE_MEM_HOST_PINNED;
    meepo::TieredTable tiered{meepo::Table(ho), meepo::Table(co)};
    tiered.hot().insert(d_keys.p, d_rows.p, n / 2);
    tiered.cold().insert(d_keys.p + n / 2, d_rows.p + (n / 2) * dim, n - n / 2);
    CHECK(tiered.size() == n);
    tiered.find(d_keys.p, n, d_out.p, d_found.p);
    HIPCK(hipDeviceSynchronize());
    { auto out = d_out.down(n * dim); auto f = d_found.down(n);
      for (size_t i = 0; i < n; ++i) CHECK(f[i] == 1);
      CHECK(memcmp(out.data(), rows.data(), n * dim * 4) == 0); }

    // ---- sharded pipeline with one rank: partition -> peer push -> peer find -> rows already in batch order ---------
    meepo::Table shard(o);
    shard.insert(d_keys.p, d_rows.p, n);
    meepo::Router router(0, n, 1);
    meepo::PeerExchange px(0, 1, 0, n, n, dim);
    char handles[MEE_P2P_BUFFERS * MEE_IPC_HANDLE_BYTES];
    px.export_handles(handles);
    px.connect(handles);
    DevBuf<int64_t> d_send(n), d_perm(n); DevBuf<uint64_t> d_counts(1);
    std::vector<int64_t> q(n);
    for (size_t i = 0; i < n; ++i) q[i] = keys[(i * 7919) % n];     // a permutation of the keys
    q[5] = (int64_t)mix64(123456789);                                // one absent key
    DevBuf<int64_t> d_q(n); d_q.up(q);
    router.partition(d_q.p, n, d_send.p, d_counts.p, d_perm.p);
    px.push(router, d_send.p, d_perm.p, d_counts.p, n);
    px.barrier();                                                    // one rank: returns at once (its own flag)
    px.find(shard);
    px.barrier();
    CHECK(px.status() == 0);
    { std::vector<float> out(n * dim); std::vector<uint8_t> f(n);
      HIPCK(hipMemcpy(out.data(), px.rows(), n * dim * 4, hipMemcpyDeviceToHost));
      HIPCK(hipMemcpy(f.data(), px.found(), n, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) {
          if (i == 5) { CHECK(f[i] == 0 && out[i * dim] == -2.0f); continue; }
          CHECK(f[i] == 1);
          CHECK(memcmp(&out[i * dim], &rows[((i * 7919) % n) * dim], dim * 4) == 0);
      } }
    // ---- mutator over the payload inbox: (key, row) pairs pushed with padding, the owner inserts its WHOLE inbox --------
    { const uint64_t slots_per_peer = n + 100;                       // > n: the segment's tail is padding (EMPTY keys)
      meepo::TableOptions so = o; so.max_batch = slots_per_peer;
      meepo::Table sink(so);
      meepo::PeerExchange pm(0, 1, 0, slots_per_peer, n, dim, /*with_payload=*/true);
      char hm[MEE_P2P_BUFFERS * MEE_IPC_HANDLE_BYTES];
      pm.export_handles(hm);
      pm.connect(hm);
      router.partition(d_keys.p, n, d_send.p, d_counts.p, d_perm.p);
      pm.push_rows(router, d_send.p, d_perm.p, d_counts.p, d_rows.p, n);
      CHECK(pm.inbox_slots() == slots_per_peer);
      sink.insert(pm.inbox_keys(), pm.inbox_rows(), pm.inbox_slots());
      CHECK(sink.size() == n && sink.status() == 0 && pm.status() == 0);   // padding is silent
      sink.find(d_keys.p, n, d_out.p, d_found.p);
      HIPCK(hipDeviceSynchronize());
      auto out = d_out.down(n * dim);
      CHECK(memcmp(out.data(), rows.data(), n * dim * 4) == 0); }
    // ---- the RCCL exchange behind the C-ABI, one rank: communicator made through the library, exact and padded layouts ----
    { char id[MEE_COMM_ID_BYTES];
      meepo::Communicator::unique_id(id);
      meepo::Communicator comm(id, 1, 0, 0);
      for (double slack : {0.0, 1.25}) {
          meepo::TableOptions so = o; so.optimizer = MEE_OPT_ADAGRAD; so.max_batch = 2 * n + 2048;
          meepo::Table shard2(so);
          meepo::ShardedTable st(shard2, comm.handle(), n, slack);
          st.insert(d_keys.p, d_rows.p, n);
          CHECK(st.size() == n);
          st.find(d_q.p, n, d_out.p, d_found.p);
          HIPCK(hipDeviceSynchronize());
          auto out = d_out.down(n * dim); auto f = d_found.down(n);
          for (size_t i = 0; i < n; ++i) {
              if (i == 5) { CHECK(f[i] == 0 && out[i * dim] == -2.0f); continue; }
              CHECK(f[i] == 1);
              CHECK(memcmp(&out[i * dim], &rows[((i * 7919) % n) * dim], dim * 4) == 0);
          }
          // one Adagrad step with zero gradients leaves the rows alone; assign of the same rows reports every key present
          HIPCK(hipMemset(d_out.p, 0, n * dim * 4));
          st.apply_adagrad(d_keys.p, d_out.p, n, 0.1f);
          st.assign(d_keys.p, d_rows.p, n, d_found.p);
          st.remove(d_q.p, 1, nullptr);                                // q[0] is a stored key
          HIPCK(hipDeviceSynchronize());
          f = d_found.down(n);
          for (size_t i = 0; i < n; ++i) CHECK(f[i] == 1);
          CHECK(st.size() == n - 1 && st.status() == 0 && shard2.status() == 0);
      } }
    // ---- the same over a hot/cold pair with pre-exchange dedup (mee_sharded_create_ex: BASELINE configs[4] behind the C-ABI), one rank -------
    { char id[MEE_COMM_ID_BYTES];
      meepo::Communicator::unique_id(id);
      meepo::Communicator comm(id, 1, 0, 0);
      meepo::TableOptions ho2 = o; ho2.optimizer = MEE_OPT_ADAGRAD; ho2.max_batch = n; ho2.capacity = (uint64_t)(n / 2);
      meepo::TableOptions co2 = ho2; co2.capacity = (uint64_t)(n / 0.75); co2.value_memory = MEE_MEM_HOST_PINNED;
      meepo::Table hot2(ho2), cold2(co2);
      meepo::ShardedTable st(hot2, comm.handle(), n, 0.0, &cold2, MEE_SHARDED_DEDUP, /*hot_key_limit=*/n / 4);
      st.insert(d_keys.p, d_rows.p, n / 4);                         // fits the hot tier
      st.insert(d_keys.p + n / 4, d_rows.p + (n / 4) * dim, n - n / 4);   // does not: goes cold
      CHECK(st.size() == n && hot2.size() == n / 4 && cold2.size() == n - n / 4);
      st.find(d_q.p, n, d_out.p, d_found.p);
      HIPCK(hipDeviceSynchronize());
      auto out = d_out.down(n * dim); auto f = d_found.down(n);
      for (size_t i = 0; i < n; ++i) {
          if (i == 5) { CHECK(f[i] == 0 && out[i * dim] == -2.0f); continue; }
          CHECK(f[i] == 1);
          CHECK(memcmp(&out[i * dim], &rows[((i * 7919) % n) * dim], dim * 4) == 0);
      }
      // a batch of ONE repeated key: de-duplicated before the exchange, every occurrence gets the row
      std::vector<int64_t> rep(n, keys[17]);
      DevBuf<int64_t> d_rep(n); d_rep.up(rep);
      st.find(d_rep.p, n, d_out.p, d_found.p);
      HIPCK(hipDeviceSynchronize());
      out = d_out.down(n * dim); f = d_found.down(n);
      for (size_t i = 0; i < n; i += 97) CHECK(f[i] == 1 && memcmp(&out[i * dim], &rows[17 * dim], dim * 4) == 0);
      st.remove(d_q.p, 1, nullptr);
      CHECK(st.size() == n - 1 && st.status() == 0 && hot2.status() == 0 && cold2.status() == 0);
    }
    printf("host_cpp_test ok: Table, TieredTable (HBM + pinned host), the peer-mapped sharded pipeline and the RCCL-backed ShardedTable through meepo_embedding.hpp\n");
    return 0;
} catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 5;
}

// meepo_embedding.hpp — C++17 host-side interface over the C-ABI (include/meepo_embedding.h), header-only.
//
// This is the "C++ host code" BASELINE.json's north_star puts above the thin C-ABI: RAII owners and lookuptable-style
// methods (find / insert / assign / remove / find_or_insert / export / size, sparse Adagrad/Adam apply), a hot/cold
// pair, the shard router and the peer-mapped exchange context.  The reference snapshot defines no C++ interface
// (/root/reference/README.md:2 is its only functional statement); names follow the verbs the north_star lists.
// Every pointer argument is DEVICE memory; every call is asynchronous on the given stream unless noted; errors are
// thrown as meepo::Error (code + the library's message).  tests/cabi/host_cpp_test.cpp drives all of it on the GPU.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>

#include "meepo_embedding.h"

namespace meepo {

class Error : public std::runtime_error {
public:
    Error(int code, const std::string& msg) : std::runtime_error("meepo error " + std::to_string(code) + ": " + msg), code_(code) {}
    int code() const noexcept { return code_; }
private:
    int code_;
};

inline void check(int rc) {
    if (rc != MEE_OK) throw Error(rc, mee_last_error());
}

struct TableOptions {
    int32_t device = 0;
    uint64_t capacity = 0;      // slots (rounded up to 16 x a prime); size for load <= 0.85
    uint32_t dim = 64;
    uint32_t optimizer = MEE_OPT_NONE;
    uint64_t max_batch = 1u << 20;
    float default_value = 0.0f;
    float initial_accumulator = 0.0f;
    uint32_t initializer = MEE_INIT_CONSTANT;
    float init_scale = 0.0f;
    uint64_t init_seed = 0;
    uint32_t value_memory = MEE_MEM_HBM;  // MEE_MEM_HOST_PINNED = cold tier (rows in pinned host DRAM)
};

// One HBM-resident shard (SPEC.md §2-§4).  Move-only.
class Table {
public:
    explicit Table(const TableOptions& o) {
        mee_config c{};
        c.struct_size = sizeof c; c.device = o.device; c.capacity = o.capacity; c.dim = o.dim; c.optimizer = o.optimizer;
        c.max_batch = o.max_batch; c.default_value = o.default_value; c.initial_accumulator = o.initial_accumulator;
        c.initializer = o.initializer; c.init_scale = o.init_scale; c.init_seed = o.init_seed; c.value_memory = o.value_memory;
        check(mee_table_create(&c, &t_));
    }
    ~Table() { if (t_) mee_table_destroy(t_); }
    Table(Table&& other) noexcept : t_(std::exchange(other.t_, nullptr)) {}
    Table& operator=(Table&& other) noexcept { if (this != &other) { if (t_) mee_table_destroy(t_); t_ = std::exchange(other.t_, nullptr); } return *this; }
    Table(const Table&) = delete;
    Table& operator=(const Table&) = delete;

    mee_table* handle() const noexcept { return t_; }
    mee_table_info info() const { mee_table_info i{}; check(mee_table_info_get(t_, &i)); return i; }

    void find(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream = nullptr) const { check(mee_find(t_, d_keys, n, d_out, d_found, stream)); }
    // ... with this call's cache policy (MEE_FIND_* flags)
    void find(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, uint32_t flags, void* stream) const { check(mee_find_ex(t_, d_keys, n, d_out, d_found, flags, stream)); }
    // embedding bag: d_out[b,:] = sum (MEE_POOL_SUM) or mean (MEE_POOL_MEAN) of the rows of d_keys[d_bag_offsets[b] .. d_bag_offsets[b+1])
    void find_pooled(const int64_t* d_keys, size_t n, const uint64_t* d_bag_offsets, size_t n_bags, float* d_out, uint8_t* d_found = nullptr, int mode = MEE_POOL_SUM, void* stream = nullptr) const {
        check(mee_find_pooled(t_, d_keys, n, d_bag_offsets, n_bags, d_out, d_found, mode, stream));
    }
    // second-tier pass: fills only the positions an earlier find (on another table) left with d_found == 0
    void find_missing(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream = nullptr) const { check(mee_find_missing(t_, d_keys, n, d_out, d_found, stream)); }
    void insert(const int64_t* d_keys, const float* d_values, size_t n, void* stream = nullptr) { check(mee_insert(t_, d_keys, d_values, n, stream)); }
    void assign(const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_assign(t_, d_keys, d_values, n, d_found, stream)); }
    void remove(const int64_t* d_keys, size_t n, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_remove(t_, d_keys, n, d_found, stream)); }
    void find_or_insert(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_find_or_insert(t_, d_keys, n, d_out, d_found, stream)); }
    // the forward of a training step over a growing vocabulary: rows + the slot of every key, for apply_*_located of the same step
    void find_or_insert_located(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream = nullptr) { check(mee_find_or_insert_located(t_, d_keys, n, d_out, d_found, d_slots_out, stream)); }
    // forward + backward of one training step: the located lookup (prepare = true: its launch also partitions the batch for the apply of the
    // SAME d_keys / n that must follow — mee_find_located_prepare / mee_find_or_insert_located_prepare), then the update at the slots it found
    void find_located(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, bool prepare = false, void* stream = nullptr) {
        check(prepare ? mee_find_located_prepare(t_, d_keys, n, d_out, d_found, d_slots_out, stream) : mee_find_located(t_, d_keys, n, d_out, d_found, d_slots_out, stream));
    }
    void find_or_insert_located_prepare(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, int64_t* d_slots_out, void* stream = nullptr) { check(mee_find_or_insert_located_prepare(t_, d_keys, n, d_out, d_found, d_slots_out, stream)); }
    void apply_adagrad_located(const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) { check(mee_apply_adagrad_located(t_, d_keys, d_slots, d_grads, n, lr, eps, stream)); }
    void apply_adam_located(const int64_t* d_keys, const int64_t* d_slots, const float* d_grads, size_t n, float lr, uint64_t step, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f, void* stream = nullptr) {
        check(mee_apply_adam_located(t_, d_keys, d_slots, d_grads, n, lr, beta1, beta2, eps, step, stream));
    }
    void apply_discard(void* stream = nullptr) { check(mee_apply_discard(t_, stream)); }
    void find_plane(uint32_t plane, const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found = nullptr, void* stream = nullptr) const { check(mee_find_plane(t_, plane, d_keys, n, d_out, d_found, stream)); }
    void assign_plane(uint32_t plane, const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_assign_plane(t_, plane, d_keys, d_values, n, d_found, stream)); }
    void apply_adagrad(const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) { check(mee_apply_adagrad(t_, d_keys, d_grads, n, lr, eps, stream)); }
    void apply_adam(const int64_t* d_keys, const float* d_grads, size_t n, float lr, uint64_t step, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f, void* stream = nullptr) {
        check(mee_apply_adam(t_, d_keys, d_grads, n, lr, beta1, beta2, eps, step, stream));
    }
    // backward of find_pooled: position i takes grad row d_grad_index[i] (its bag)
    void apply_adagrad_indexed(const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) {
        check(mee_apply_adagrad_indexed(t_, d_keys, d_grads, n_grad_rows, d_grad_index, n, lr, eps, stream));
    }
    void apply_adam_indexed(const int64_t* d_keys, const float* d_grads, size_t n_grad_rows, const uint32_t* d_grad_index, size_t n, float lr, uint64_t step, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f, void* stream = nullptr) {
        check(mee_apply_adam_indexed(t_, d_keys, d_grads, n_grad_rows, d_grad_index, n, lr, beta1, beta2, eps, step, stream));
    }
    // the next four synchronise the stream (they return host values)
    size_t size(void* stream = nullptr) const { size_t n = 0; check(mee_size(t_, &n, stream)); return n; }
    uint32_t status(void* stream = nullptr) const { uint32_t b = 0; check(mee_status(t_, &b, stream)); return b; }
    size_t export_all(int64_t* d_keys_out, float* d_values_out, size_t cap, float* d_state1_out = nullptr, float* d_state2_out = nullptr, void* stream = nullptr) const {
        size_t n = 0; check(mee_export(t_, d_keys_out, d_values_out, d_state1_out, d_state2_out, cap, &n, stream)); return n;
    }
    // pairs stored in slots [slot_begin, slot_end): checkpoint / rehash in bounded pieces; returns how many the range holds
    size_t export_range(uint64_t slot_begin, uint64_t slot_end, int64_t* d_keys_out, float* d_values_out, size_t cap, float* d_state1_out = nullptr, float* d_state2_out = nullptr, void* stream = nullptr) const {
        size_t n = 0; check(mee_export_range(t_, slot_begin, slot_end, d_keys_out, d_values_out, d_state1_out, d_state2_out, cap, &n, stream)); return n;
    }
    // sync-free: padded outputs of length n (MEE_EMPTY_KEY between the distinct keys, counts 0 there); see mee_dedup_sum
    void dedup_sum(const int64_t* d_keys, const float* d_grads, size_t n, int64_t* d_uniq_out, float* d_gsum_out, uint32_t* d_counts_out, int64_t* d_inverse_out,
                   int64_t miss_index = -1, void* stream = nullptr) {
        check(mee_dedup_sum(t_, d_keys, d_grads, n, d_uniq_out, d_gsum_out, d_counts_out, d_inverse_out, miss_index, stream));
    }
    // sync-free: d_uniq_out[n] = every distinct key once, EMPTY everywhere else (also between the keys), d_inverse_out[i] = index into it (miss_index for reserved keys)
    void dedup_keys(const int64_t* d_keys, size_t n, int64_t* d_uniq_out, int64_t* d_inverse_out, int64_t miss_index = -1, void* stream = nullptr) {
        check(mee_dedup_keys(t_, d_keys, n, d_uniq_out, d_inverse_out, miss_index, stream));
    }
    void clear(void* stream = nullptr) { check(mee_clear(t_, stream)); }
    // rehash in place to at least `capacity` slots (synchronises; old and new planes must fit together)
    void reserve(uint64_t capacity, void* stream = nullptr) { check(mee_reserve(t_, capacity, stream)); }
    void clear_status(void* stream = nullptr) { check(mee_clear_status(t_, stream)); }
    void set_tuning(const char* name, int value) { check(mee_set_tuning(t_, name, value)); }

private:
    mee_table* t_ = nullptr;
};

// Many tables (same device, same dim) served by ONE find launch over their concatenated ("jagged") key batches.
class Group {
public:
    Group(Table* const* tables, uint32_t n, uint64_t max_apply_batch = 0) {
        std::string h(n * sizeof(mee_table*), '\0');
        auto** raw = reinterpret_cast<mee_table**>(&h[0]);
        for (uint32_t j = 0; j < n; ++j) raw[j] = tables[j]->handle();
        check(mee_group_create(raw, n, max_apply_batch, &g_));
    }
    ~Group() { if (g_) mee_group_destroy(g_); }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    // segment j = d_keys[d_offsets[j] .. d_offsets[j+1]); d_offsets: n_tables + 1 values in DEVICE memory; n = total positions
    void find(const int64_t* d_keys, const uint64_t* d_offsets, size_t n, float* d_out, uint8_t* d_found, void* stream = nullptr) {
        check(mee_find_grouped(g_, d_keys, d_offsets, n, d_out, d_found, stream));
    }
    void set_tuning(const char* name, int value) { check(mee_group_set_tuning(g_, name, value)); }   // knobs of the group's own apply; never change results
    void apply_adagrad(const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) {
        check(mee_group_apply_adagrad(g_, d_keys, d_offsets, d_grads, n, lr, eps, stream));
    }
    void apply_adam(const int64_t* d_keys, const uint64_t* d_offsets, const float* d_grads, size_t n, float lr, uint64_t step, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f, void* stream = nullptr) {
        check(mee_group_apply_adam(g_, d_keys, d_offsets, d_grads, n, lr, beta1, beta2, eps, step, stream));
    }
private:
    mee_group* g_ = nullptr;
};

// Hot (HBM) table backed by a cold table whose rows live in pinned host DRAM: one logical table, sync-free lookup.
class TieredTable {
public:
    TieredTable(Table hot, Table cold) : hot_(std::move(hot)), cold_(std::move(cold)) {}
    Table& hot() noexcept { return hot_; }
    Table& cold() noexcept { return cold_; }
    void find(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found, void* stream = nullptr) const {
        hot_.find(d_keys, n, d_out, d_found, stream);          // d_found is required: the second pass reads it
        cold_.find_missing(d_keys, n, d_out, d_found, stream);
    }
    void apply_adagrad(const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) {
        hot_.apply_adagrad(d_keys, d_grads, n, lr, eps, stream);  // a key lives in one tier; each table ignores keys it does not hold
        cold_.apply_adagrad(d_keys, d_grads, n, lr, eps, stream);
    }
    size_t size(void* stream = nullptr) const { return hot_.size(stream) + cold_.size(stream); }
private:
    Table hot_, cold_;
};

// Shard partition / un-permute workspace (SPEC.md §5).
class Router {
public:
    Router(int32_t device, uint64_t max_batch, uint32_t n_shards) { check(mee_router_create(device, max_batch, n_shards, &r_)); }
    ~Router() { if (r_) mee_router_destroy(r_); }
    Router(Router&& o) noexcept : r_(std::exchange(o.r_, nullptr)) {}
    Router(const Router&) = delete;
    Router& operator=(const Router&) = delete;
    mee_router* handle() const noexcept { return r_; }
    void partition(const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm, void* stream = nullptr) { check(mee_partition(r_, d_keys, n, d_send_keys, d_counts, d_perm, stream)); }
    // the same, EMPTY keys (padding) belong to no shard
    void partition_padded(const int64_t* d_keys, size_t n, int64_t* d_send_keys, uint64_t* d_counts, int64_t* d_perm, void* stream = nullptr) { check(mee_partition_padded(r_, d_keys, n, d_send_keys, d_counts, d_perm, stream)); }
    static void scatter_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream = nullptr) { check(mee_scatter_rows(d_rows, d_perm, n, row_bytes, d_out, stream)); }
    static void gather_rows(const void* d_rows, const int64_t* d_perm, size_t n, size_t row_bytes, void* d_out, void* stream = nullptr) { check(mee_gather_rows(d_rows, d_perm, n, row_bytes, d_out, stream)); }
private:
    mee_router* r_ = nullptr;
};

// Peer-mapped exchange context of the all-to-all-free sharded find.  The caller moves the exported handles between
// ranks (any side channel); the two cross-rank barriers per lookup are barrier() (or any collective on the stream):
// partition -> push -> barrier -> find -> barrier -> read rows()/found().
class PeerExchange {
public:
    PeerExchange(int32_t device, uint32_t n_shards, uint32_t rank, uint64_t slots_per_peer, uint64_t max_batch, uint32_t dim, bool with_payload = false) {
        check(mee_p2p_create(device, n_shards, rank, slots_per_peer, max_batch, dim, with_payload ? 1 : 0, &c_));
    }
    ~PeerExchange() { if (c_) mee_p2p_destroy(c_); }
    PeerExchange(const PeerExchange&) = delete;
    PeerExchange& operator=(const PeerExchange&) = delete;
    void export_handles(void* handles_6x64) { check(mee_p2p_export(c_, handles_6x64)); }
    void connect(const void* all_handles_rank_major) { check(mee_p2p_connect(c_, all_handles_rank_major)); }
    void push(Router& r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts, size_t n, void* stream = nullptr) { check(mee_p2p_push(c_, r.handle(), d_send_keys, d_perm, d_counts, n, stream)); }
    void find(const Table& t, void* stream = nullptr) { check(mee_p2p_find(c_, t.handle(), stream)); }
    // the cross-rank barrier of the sequences above as a one-wave kernel over peer-mapped flags (no collective library needed)
    void barrier(void* stream = nullptr) { check(mee_p2p_barrier(c_, stream)); }
    void push_rows(Router& r, const int64_t* d_send_keys, const int64_t* d_perm, const uint64_t* d_counts, const float* d_rows, size_t n, void* stream = nullptr) {
        check(mee_p2p_push_rows(c_, r.handle(), d_send_keys, d_perm, d_counts, d_rows, n, stream));
    }
    int64_t* inbox_keys() const { int64_t* k = nullptr; check(mee_p2p_inbox(c_, &k, nullptr, nullptr)); return k; }
    float* inbox_rows() const { float* r = nullptr; check(mee_p2p_inbox(c_, nullptr, &r, nullptr)); return r; }
    uint64_t inbox_slots() const { uint64_t n = 0; check(mee_p2p_inbox(c_, nullptr, nullptr, &n)); return n; }
    float* rows() const { float* o = nullptr; check(mee_p2p_buffers(c_, &o, nullptr)); return o; }
    uint8_t* found() const { uint8_t* f = nullptr; check(mee_p2p_buffers(c_, nullptr, &f)); return f; }
    uint32_t status(void* stream = nullptr) { uint32_t b = 0; check(mee_p2p_status(c_, &b, stream)); return b; }
private:
    mee_p2p* c_ = nullptr;
};

// An RCCL communicator made through the library (callers that link RCCL themselves pass their own ncclComm_t instead).
class Communicator {
public:
    static void unique_id(void* id_128_bytes) { check(mee_comm_unique_id(id_128_bytes)); }   // one rank calls, all ranks share the bytes
    Communicator(const void* id_128_bytes, uint32_t n_ranks, uint32_t rank, int32_t device) { check(mee_comm_create(id_128_bytes, n_ranks, rank, device, &c_)); }
    ~Communicator() { if (c_) mee_comm_destroy(c_); }
    Communicator(const Communicator&) = delete;
    Communicator& operator=(const Communicator&) = delete;
    void* handle() const noexcept { return c_; }   // an ncclComm_t
private:
    void* c_ = nullptr;
};

// One rank's view of the row-sharded table (SPEC.md §5): every verb is collective over the communicator's ranks and runs the
// whole exchange — partition, grouped ncclSend/ncclRecv, the local shard's operator, rows back, un-permute — on `stream`.
class ShardedTable {
public:
    // pad_slack = 0: exact message sizes (one host sync per call); >= 1: fixed EMPTY-padded segments, no host sync
    ShardedTable(Table& local, void* nccl_comm, uint64_t max_batch, double pad_slack = 0.0) { check(mee_sharded_create(local.handle(), nccl_comm, max_batch, pad_slack, &s_)); }
    // with options: a cold table behind `local` (every shard a hot/cold pair), pre-exchange dedup of lookups (MEE_SHARDED_DEDUP)
    ShardedTable(Table& local, void* nccl_comm, uint64_t max_batch, double pad_slack, Table* cold, uint32_t flags = 0, uint64_t hot_key_limit = 0) {
        mee_sharded_options o{};
        o.struct_size = sizeof o; o.flags = flags; o.max_batch = max_batch; o.pad_slack = pad_slack; o.cold = cold ? cold->handle() : nullptr; o.hot_key_limit = hot_key_limit;
        check(mee_sharded_create_ex(local.handle(), nccl_comm, &o, &s_));
    }
    ~ShardedTable() { if (s_) mee_sharded_destroy(s_); }
    ShardedTable(const ShardedTable&) = delete;
    ShardedTable& operator=(const ShardedTable&) = delete;
    void find(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_sharded_find(s_, d_keys, n, d_out, d_found, stream)); }
    void find_or_insert(const int64_t* d_keys, size_t n, float* d_out, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_sharded_find_or_insert(s_, d_keys, n, d_out, d_found, stream)); }
    void insert(const int64_t* d_keys, const float* d_values, size_t n, void* stream = nullptr) { check(mee_sharded_insert(s_, d_keys, d_values, n, stream)); }
    void assign(const int64_t* d_keys, const float* d_values, size_t n, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_sharded_assign(s_, d_keys, d_values, n, d_found, stream)); }
    void remove(const int64_t* d_keys, size_t n, uint8_t* d_found = nullptr, void* stream = nullptr) { check(mee_sharded_remove(s_, d_keys, n, d_found, stream)); }
    void apply_adagrad(const int64_t* d_keys, const float* d_grads, size_t n, float lr, float eps = 1e-10f, void* stream = nullptr) { check(mee_sharded_apply_adagrad(s_, d_keys, d_grads, n, lr, eps, stream)); }
    void apply_adam(const int64_t* d_keys, const float* d_grads, size_t n, float lr, float b1, float b2, float eps, uint64_t step, void* stream = nullptr) { check(mee_sharded_apply_adam(s_, d_keys, d_grads, n, lr, b1, b2, eps, step, stream)); }
    size_t size(void* stream = nullptr) { size_t n = 0; check(mee_sharded_size(s_, &n, stream)); return n; }
    uint32_t status(void* stream = nullptr) { uint32_t b = 0; check(mee_sharded_status(s_, &b, stream)); return b; }
    void clear_status(void* stream = nullptr) { check(mee_sharded_clear_status(s_, stream)); }
private:
    mee_sharded* s_ = nullptr;
};

}  // namespace meepo

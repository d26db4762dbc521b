"""CPU: pin the oracle — hashing KATs (two implementations + committed fixture), dict-model semantics,
optimizer math vs torch.optim golden vectors.  The reference has no vectors of its own (parity unpinned)."""
import json
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle
from meepoembedding_amd import synth
from oracle import pyspec

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "hash_kat.json")) as f:
        return json.load(f)


def test_hash_kat_c_oracle(kat):
    keys = np.array([int(k) for k in kat["keys"]], dtype=np.int64)
    for nb, exp_b in kat["bucket"].items():
        for g, exp_o in kat["owner"].items():
            mix, bkt, own = oracle.hash_batch(keys, int(nb), int(g))
            assert [str(int(x)) for x in mix] == kat["mix64"]
            assert [str(int(x)) for x in bkt] == exp_b
            assert [int(x) for x in own] == exp_o


def test_hash_kat_pyspec(kat):
    for i, k in enumerate(kat["keys"]):
        k = int(k)
        assert str(pyspec.mix64(pyspec.u64(k))) == kat["mix64"][i]
        assert str(pyspec.mix64b(pyspec.u64(k))) == kat["mix64b"][i]
        for nb, exp in kat["bucket"].items():
            assert str(pyspec.bucket(k, int(nb))) == exp[i]
        for g, exp in kat["owner"].items():
            assert pyspec.owner(k, int(g)) == exp[i]
        for nb, exp in kat["step"].items():
            assert str(pyspec.step(k, int(nb))) == exp[i] and str(oracle.lib().meo_step(k, int(nb))) == exp[i]
            assert 1 <= int(exp[i]) < max(2, int(nb))


def published_kat():
    with open(os.path.join(GOLDEN, "published_kat.json")) as f:
        return json.load(f)


def splitmix_inputs_outputs(pub):
    """(mix64 input, published output) pairs: output k of the stream seeded with s is mix64(s + (k + 1) * gamma)."""
    gamma = int(pub["splitmix64"]["gamma"], 0)
    pairs = []
    for st_ in pub["splitmix64"]["streams"]:
        x = int(st_["seed"], 0)
        for o in st_["outputs"]:
            x = (x + gamma) & ((1 << 64) - 1)
            pairs.append((x, int(o, 0)))
    return pairs


def test_published_known_answers():
    """Third-party anchors for the integer half of the oracle: the PUBLISHED outputs of splitmix64 (SPEC.md mix64) and of murmur3's fmix64
    (SPEC.md mix64b), against the pure-Python restatement AND the C oracle."""
    pub = published_kat()
    L = oracle.lib()
    as_i64 = lambda x: x - (1 << 64) if x >= 1 << 63 else x
    for x, want in splitmix_inputs_outputs(pub):
        assert pyspec.mix64(x) == want
        mix, _, _ = oracle.hash_batch(np.array([as_i64(x)], dtype=np.int64), 1, 1)
        assert int(mix[0]) == want
    for a, b in pub["fmix64"]["pairs"]:
        assert pyspec.mix64b(int(a, 0)) == int(b, 0)
        assert int(L.meo_mix64b(int(a, 0))) == int(b, 0)


def test_hash_ranges(kat):
    keys = np.array([int(k) for k in kat["keys"]], dtype=np.int64)
    _, bkt, own = oracle.hash_batch(keys, 625, 8)
    assert bkt.max() < 625 and own.max() < 8
    # owner roughly balanced on a larger stream
    _, _, own = oracle.hash_batch(synth.keys_np(3, 0, 80000), 1, 8)
    cnt = np.bincount(own, minlength=8)
    assert cnt.min() > 9000 and cnt.max() < 11000


def test_initial_row_kat(kat):
    ir = kat["initial_row"]
    t = oracle.OracleTable(64, ir["dim"], initializer=oracle.INIT_UNIFORM, init_scale=ir["scale"], init_seed=ir["seed"])
    for k, row in zip(kat["keys"], ir["rows"]):
        got = t.initial_row(int(k))
        assert np.array_equal(got, np.array(row, np.float32))
        assert np.all(np.abs(got) <= ir["scale"])


def test_synth_generators_agree():
    import torch
    k = synth.keys_np(1, 10, 500)
    assert np.array_equal(k, synth.keys_t(1, 10, 500, "cpu").numpy())
    assert k[0] == pyspec.splitmix64_stream(1, 10)
    assert np.array_equal(synth.rows_np(k, 16, 2), synth.rows_t(torch.from_numpy(k), 16, 2).numpy())


def test_config0_roundtrip_cpu():
    """BASELINE configs[0]: 1M int64 keys, dim 16, find/insert/assign/export round trip on the CPU backend."""
    n, dim = 1_000_000, 16
    keys = synth.keys_np(1, 0, n)
    rows = synth.rows_np(keys, dim, 2)
    t = oracle.OracleTable(int(n / 0.75), dim)
    t.insert(keys, rows)
    assert t.size() == n and t.status() == 0
    out, found = t.find(keys)
    assert found.all() and np.array_equal(out, rows)
    absent = synth.keys_np(99, 0, n)
    out, found = t.find(absent)
    assert not found.any() and not out.any()
    sub = keys[::2]
    new = synth.rows_np(sub, dim, 5)
    assert t.assign(sub, new).all()
    assert not t.assign(absent[:1000], new[:1000]).any()
    out, _ = t.find(keys)
    assert np.array_equal(out[::2], new) and np.array_equal(out[1::2], rows[1::2])
    ek, ev = t.export()
    order = np.argsort(ek)
    ref = np.argsort(keys)
    assert np.array_equal(ek[order], keys[ref])
    assert np.array_equal(ev[order], out[ref])


def test_reserved_keys_and_table_full():
    t = oracle.OracleTable(32, 4)
    keys = np.array([oracle.EMPTY_KEY, 5, oracle.RECLAIMED_KEY, 6], dtype=np.int64)
    t.insert(keys, np.ones((4, 4), np.float32))
    assert t.size() == 2 and t.status() == oracle.STATUS_RESERVED_KEY
    t2 = oracle.OracleTable(32, 4)
    t2.insert(np.array([oracle.EMPTY_KEY, 9], dtype=np.int64), np.ones((2, 4), np.float32))
    assert t2.size() == 1 and t2.status() == 0, "EMPTY in a batch is padding: skipped silently"
    out, found = t.find(keys)
    assert list(found) == [0, 1, 0, 1]
    t.clear_status()
    assert t.capacity == 32                 # 2 buckets: 2 is prime
    many = synth.keys_np(4, 0, 100)
    t.insert(many, np.zeros((100, 4), np.float32))
    assert t.size() == 32 and t.status() & oracle.STATUS_TABLE_FULL
    # every stored key is still findable, every dropped key is absent
    _, found = t.find(many)
    assert found.sum() == 30


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.lists(st.tuples(st.sampled_from(["insert", "assign", "find", "remove", "insert"]),
                          st.lists(st.integers(min_value=-40, max_value=40), min_size=0, max_size=30)),
                min_size=1, max_size=12),
       st.sampled_from([16, 48, 96]))
def test_model_vs_dict(ops, capacity):
    """SPEC §2-§3 against a Python dict model incl. last-wins duplicates and the table-full rule."""
    dim = 4
    t = oracle.OracleTable(capacity, dim, default_value=-1.0)
    m = pyspec.DictTable(capacity, dim, default_value=-1.0)
    ctr = 0
    for op, ks in ops:
        rows = [[float(ctr + i), 1.0, 2.0, float(k)] for i, k in enumerate(ks)]
        ctr += len(ks)
        ka = np.array(ks, dtype=np.int64)
        ra = np.array(rows, dtype=np.float32).reshape(len(ks), dim)
        if op == "insert":
            t.insert(ka, ra); m.insert(ks, rows)
        elif op == "assign":
            got = t.assign(ka, ra); exp = m.assign(ks, rows)
            assert list(got.astype(bool)) == exp
        elif op == "remove":
            got = t.remove(ka); exp = m.remove(ks)
            assert list(got.astype(bool)) == exp
        else:
            out, found = t.find(ka); eo, ef = m.find(ks)
            assert list(found.astype(bool)) == ef
            assert np.array_equal(out, np.array(eo, np.float32).reshape(len(ks), dim))
        assert t.size() == m.size()
    ek, ev = t.export()
    order = np.argsort(ek)
    mk, mv = m.export_sorted()
    assert list(ek[order]) == mk
    assert np.array_equal(ev[order], np.array(mv, np.float32).reshape(len(mk), dim))


def test_remove_and_slot_reuse_cpu():
    """remove -> RECLAIMED tombstones; later inserts reuse them, so a full table can churn forever."""
    dim = 4
    t = oracle.OracleTable(70, dim)
    cap = t.capacity                        # 5 buckets (prime) x 16 = 80 slots
    assert cap == 80
    keys = synth.keys_np(6, 0, cap)
    t.insert(keys, synth.rows_np(keys, dim, 1))
    assert t.size() == cap and t.status() == 0     # prime bucket count: every stride reaches every bucket, 100 % fill works
    for rnd in range(5):
        old = keys[rnd * 8:(rnd + 1) * 8]
        assert t.remove(old).all() and not t.remove(old).any()
        assert t.size() == cap - 8 and not t.find(old)[1].any()
        new = synth.keys_np(100 + rnd, 0, 8)
        t.insert(new, synth.rows_np(new, dim, 2))
        assert t.size() == cap and t.status() == 0, "tombstones must be reused"
        out, found = t.find(new)
        assert found.all() and np.array_equal(out, synth.rows_np(new, dim, 2))
    rest = keys[40:]
    out, found = t.find(rest)
    assert found.all() and np.array_equal(out, synth.rows_np(rest, dim, 1))


def test_find_or_insert_cpu():
    t = oracle.OracleTable(256, 8, initializer=oracle.INIT_UNIFORM, init_scale=0.1, init_seed=3, optimizer=oracle.OPT_ADAGRAD,
                           initial_accumulator=0.5)
    keys = np.array([1, 2, 1, 3, 2, 9], dtype=np.int64)
    out, found = t.find_or_insert(keys)
    assert not found.any() and t.size() == 4
    for i, k in enumerate(keys):
        assert np.array_equal(out[i], t.initial_row(int(k)))
    out2, found2 = t.find_or_insert(keys)
    assert found2.all() and np.array_equal(out, out2)
    _, _, acc, _ = t.export(with_state=True)
    assert np.all(acc == 0.5)


def test_dedup_sum_cpu():
    keys = np.array([5, 7, 5, 5, 9, 7, oracle.EMPTY_KEY], dtype=np.int64)
    g = np.arange(7 * 4, dtype=np.float32).reshape(7, 4)
    uniq, gs, inv, cnt = oracle.dedup_sum(keys, g, 4)
    assert list(uniq) == [5, 7, 9] and list(cnt) == [3, 2, 1] and list(inv) == [0, 1, 0, 0, 2, 1, -1]
    assert np.array_equal(gs[0], g[0] + g[2] + g[3]) and np.array_equal(gs[2], g[4])


def test_partition_cpu():
    keys = synth.keys_np(8, 0, 5000)
    for g in (1, 2, 8):
        send, counts, perm = oracle.partition(keys, g)
        assert counts.sum() == 5000 and np.array_equal(send, keys[perm])
        own = oracle.hash_batch(send, 1, g)[2]
        off = 0
        for p in range(g):
            seg = slice(off, off + counts[p])
            assert (own[seg] == p).all() and (np.diff(perm[seg]) > 0).all()  # stable
            off += counts[p]


@pytest.mark.parametrize("dim", [16, 64])
def test_optimizers_vs_torch_golden(dim):
    """SPEC §4 vs torch.optim.Adagrad / SparseAdam golden vectors (third-party math, duplicates coalesced)."""
    z = np.load(os.path.join(GOLDEN, "optimizer_golden.npz"))
    w0, idx, grads = z[f"w0_{dim}"], z[f"idx_{dim}"], z[f"grads_{dim}"]
    rows = w0.shape[0]
    keys = synth.keys_np(21, 0, rows)  # row r <-> key keys[r]
    for name in ("adagrad", "adam"):
        opt = oracle.OPT_ADAGRAD if name == "adagrad" else oracle.OPT_ADAM
        t = oracle.OracleTable(1024, dim, optimizer=opt, initial_accumulator=0.1 if name == "adagrad" else 0.0)
        t.insert(keys, w0)
        for s in range(idx.shape[0]):
            if name == "adagrad":
                t.apply_adagrad(keys[idx[s]], grads[s], 0.05, 1e-10)
            else:
                t.apply_adam(keys[idx[s]], grads[s], 0.01, 0.9, 0.999, 1e-8, s + 1)
        got, found = t.find(keys)
        assert found.all()
        np.testing.assert_allclose(got, z[f"{name}_w_{dim}"], rtol=2e-6, atol=1e-7)


def golden_cases():
    z = np.load(os.path.join(GOLDEN, "optimizer_golden.npz"))
    return z, json.loads(str(z["cases_json"]))


def test_optimizers_vs_torch_golden_more_cases():
    """Round 4's cases: dim 128, non-default eps / betas, initial accumulator 0 and 0.1 (tests/golden/make_golden.py)."""
    z, cases = golden_cases()
    assert len(cases) >= 5
    for c in cases:
        t_ = c["tag"]
        w0, idx, grads = z[f"c_{t_}_w0"], z[f"c_{t_}_idx"], z[f"c_{t_}_grads"]
        keys = synth.keys_np(22, 0, w0.shape[0])
        adagrad = c["opt"] == "adagrad"
        t = oracle.OracleTable(1024, c["dim"], optimizer=oracle.OPT_ADAGRAD if adagrad else oracle.OPT_ADAM, initial_accumulator=c.get("acc0", 0.0))
        t.insert(keys, w0)
        for s in range(idx.shape[0]):
            if adagrad:
                t.apply_adagrad(keys[idx[s]], grads[s], c["lr"], c["eps"])
            else:
                t.apply_adam(keys[idx[s]], grads[s], c["lr"], c["beta1"], c["beta2"], c["eps"], s + 1)
        got, found = t.find(keys)
        assert found.all()
        np.testing.assert_allclose(got, z[f"c_{t_}_w"], rtol=2e-6, atol=1e-7, err_msg=t_)


def test_pool_rows_known_answer():
    """oracle.pool_rows: position-order fp32 sums per bag (hand-checked), empty bags are zeros, mean divides by the length."""
    rows = np.array([[1.0, 1e8], [2.0, 1.0], [3.0, -1e8], [4.0, 0.5], [5.0, 0.25]], dtype=np.float32)
    off = np.array([0, 3, 3, 4, 5])
    s = oracle.pool_rows(rows, off, "sum")
    # (1e8 + 1) rounds to 1e8 in fp32, then -1e8 gives 0: the ORDER is part of the definition
    assert s.tolist() == [[6.0, 0.0], [0.0, 0.0], [4.0, 0.5], [5.0, 0.25]]
    m = oracle.pool_rows(rows, off, "mean")
    assert m.tolist() == [[2.0, 0.0], [0.0, 0.0], [4.0, 0.5], [5.0, 0.25]]
    assert oracle.pool_rows(rows[:0], np.array([0]), "sum").shape == (0, 2)

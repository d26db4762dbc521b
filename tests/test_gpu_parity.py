"""GPU parity: every operator of the HIP backend through the C-ABI vs the CPU oracle on the same seeded inputs.
Bar: bit-exact for hashing / found-masks / copied rows / size / key-sorted export; ≤1e-6 relative (atol 1e-9) for
fp32 optimizer state (SPEC.md §4)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from meepoembedding_amd import _lib
from meepoembedding_amd import (INIT_UNIFORM, OPT_ADAGRAD, OPT_ADAM, STATUS_RESERVED_KEY, STATUS_TABLE_FULL, LookupTable,
                                MeepoError, Router, hash_batch, synth)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RTOL, ATOL = 1e-6, 1e-9


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def sorted_export(keys, *planes):
    order = np.argsort(keys)
    return (keys[order],) + tuple(p[order] for p in planes)


def test_hash_kat_on_device(dev):
    kat = json.load(open(os.path.join(GOLDEN, "hash_kat.json")))
    keys = np.array([int(k) for k in kat["keys"]], dtype=np.int64)
    for nb, exp_b in kat["bucket"].items():
        for g, exp_o in kat["owner"].items():
            mix, bkt, own = hash_batch(T(keys, dev), int(nb), int(g))
            assert [str(int(x)) for x in mix.cpu().numpy().view(np.uint64)] == kat["mix64"]
            assert [str(int(x)) for x in bkt.cpu().numpy().view(np.uint64)] == exp_b
            assert [int(x) for x in own.cpu().numpy()] == exp_o


def test_published_known_answers_on_device(dev):
    """The PUBLISHED outputs of splitmix64 and murmur3 fmix64 (tests/golden/published_kat.json) from the device's own mixers: mix64 in full through
    mee_hash_batch's mix output, mix64b through the owner function at 2^31 shards (owner = the top 31 bits of mix64b)."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "published_kat.json")) as f:
        pub = json.load(f)
    as_i64 = lambda x: x - (1 << 64) if x >= 1 << 63 else x
    pairs, gamma = [], int(pub["splitmix64"]["gamma"], 0)   # output k of the stream seeded with s is mix64(s + (k + 1) * gamma)
    for st_ in pub["splitmix64"]["streams"]:
        x = int(st_["seed"], 0)
        for o in st_["outputs"]:
            x = (x + gamma) & ((1 << 64) - 1)
            pairs.append((x, int(o, 0)))
    mix, _, _ = hash_batch(T(np.array([as_i64(x) for x, _ in pairs], dtype=np.int64), dev), 1, 1)
    assert [int(v) for v in mix.cpu().numpy().view(np.uint64)] == [w for _, w in pairs]
    fm = pub["fmix64"]["pairs"]
    _, _, own = hash_batch(T(np.array([as_i64(int(a, 0)) for a, _ in fm if int(a, 0) != (1 << 63)], dtype=np.int64), dev), 1, 1 << 31)
    assert [int(v) for v in own.cpu().numpy()] == [int(b, 0) >> 33 for _, b in fm]


@pytest.mark.parametrize("dim,n,load", [(16, 20000, 0.75), (64, 50000, 0.75), (128, 8192, 0.9), (4, 1000, 0.5), (40, 3000, 0.75),
                                        (256, 2000, 0.75)])
def test_insert_find_assign_export(dev, dim, n, load):
    keys = synth.keys_np(1, 0, 2 * n)
    present, absent = keys[:n], keys[n:]
    rows = synth.rows_np(present, dim, 2)
    cap = int(n / load)
    t = LookupTable(cap, dim, device=dev, max_batch=2 * n, default_value=-3.5)
    o = oracle.OracleTable(cap, dim, default_value=-3.5)
    assert t.capacity == o.capacity
    t.insert(T(present, dev), T(rows, dev)); o.insert(present, rows)
    assert t.size() == o.size() == n and t.status() == 0
    # mixed hit/miss lookup, shuffled
    rng = np.random.default_rng(5)
    q = np.concatenate([present, absent])[rng.permutation(2 * n)]
    out, found = t.find(T(q, dev))
    eo, ef = o.find(q)
    assert np.array_equal(found.cpu().numpy(), ef)
    assert np.array_equal(out.cpu().numpy(), eo)
    # assign half (with misses mixed in)
    sub = np.concatenate([present[::2], absent[:100]])
    new = synth.rows_np(sub, dim, 9)
    fa = t.assign(T(sub, dev), T(new, dev)); fo = o.assign(sub, new)
    assert np.array_equal(fa.cpu().numpy(), fo)
    out, found = t.find(T(q, dev)); eo, ef = o.find(q)
    assert np.array_equal(out.cpu().numpy(), eo) and np.array_equal(found.cpu().numpy(), ef)
    gk, gv = t.export(); ok, ov = o.export()
    gk, gv = sorted_export(gk.cpu().numpy(), gv.cpu().numpy()); ok, ov = sorted_export(ok, ov)
    assert np.array_equal(gk, ok) and np.array_equal(gv, ov)
    t.clear()
    assert t.size() == 0 and not t.find(T(q[:100], dev))[1].any()


@pytest.mark.parametrize("dim", [64, 128, 24])
def test_find_many_equals_separate_finds(dev, dim):
    """mee_find_many: several requests of one table in one launch == mee_find per request (ragged sizes, an empty request, absent and
    reserved keys, a request that is not a multiple of the wave step)."""
    n_keys = 30000
    keys = synth.keys_np(17, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    t = LookupTable(65536, dim, device=dev, max_batch=n_keys, default_value=0.25)
    t.insert(T(keys, dev), T(rows, dev))
    rng = np.random.default_rng(4)
    reqs = []
    for n in (4096, 1, 0, 777, 20000, 33):
        q = keys[rng.integers(0, n_keys, n)]
        if n > 10:
            q[::7] = synth.keys_np(99, 0, q[::7].size)           # absent
            q[3] = oracle.EMPTY_KEY
        reqs.append(T(q, dev))
    got = t.find_many(reqs)
    for q, (o, f) in zip(reqs, got):
        eo, ef = t.find(q)
        assert torch.equal(o, eo) and torch.equal(f, ef)
    # preallocated outputs, 16 requests (the maximum), and one more than that is refused
    many = [(reqs[0][i * 100:(i + 1) * 100], torch.empty(100, dim, device=dev), torch.empty(100, dtype=torch.uint8, device=dev)) for i in range(16)]
    got = t.find_many(many)
    eo, ef = t.find(reqs[0][:1600])
    assert torch.equal(torch.cat([o for o, _ in got]), eo) and torch.equal(torch.cat([f for _, f in got]), ef)
    with pytest.raises(MeepoError):
        t.find_many([reqs[1]] * 17)


def test_find_unordered_same_results(dev):
    """mee_find_unordered (hipExtAnyOrderLaunch): many lookups queued back to back on one stream, each with buffers of its own, may overlap
    each other; every one of them returns what mee_find returns, and later in-order work on the stream sees the results."""
    dim, n_keys, nb = 64, 200_000, 12
    keys = synth.keys_t(23, 0, n_keys, dev)
    t = LookupTable(int(n_keys / 0.75), dim, device=dev, max_batch=n_keys)
    t.insert(keys, synth.rows_t(keys, dim, 2))
    g = torch.Generator(device=dev); g.manual_seed(3)
    batches = [keys[torch.randint(0, n_keys, (50_000,), device=dev, generator=g)] for _ in range(nb)]
    for b in batches:
        b[::9] = synth.keys_t(99, 0, b[::9].numel(), dev)    # absent
    outs = [torch.full((50_000, dim), 7.0, device=dev) for _ in range(nb)]
    founds = [torch.full((50_000,), 9, dtype=torch.uint8, device=dev) for _ in range(nb)]
    torch.cuda.synchronize(dev)                               # inputs complete, nothing earlier touches the outputs
    for _ in range(3):
        for b, o, f in zip(batches, outs, founds):
            t.find(b, out=o, found=f, unordered=True)
    sums = [o.sum() for o in outs]                            # in-order work behind the unordered launches
    torch.cuda.synchronize(dev)
    for b, o, f, s_ in zip(batches, outs, founds, sums):
        eo, ef = t.find(b)
        assert torch.equal(o, eo) and torch.equal(f, ef) and torch.equal(s_, eo.sum())


def test_find_edge_cases(dev):
    t = LookupTable(1000, 64, device=dev, max_batch=4096)
    out, found = t.find(torch.empty(0, dtype=torch.int64, device=dev))   # empty batch
    assert out.shape == (0, 64) and found.numel() == 0
    out, found = t.find(T(np.array([1, 2, 3], np.int64), dev))            # empty table, ragged (n % 4 != 0)
    assert not found.any() and not out.any()
    out, _ = t.find(T(np.array([7], np.int64), dev), want_found=False)     # found mask optional
    assert out.shape == (1, 64)
    t.insert(T(np.array([7], np.int64), dev), torch.ones(1, 64, device=dev))
    for n in (1, 2, 3, 5, 63, 64, 65, 257):
        q = np.full(n, 7, np.int64)
        out, found = t.find(T(q, dev))
        assert found.all() and (out == 1).all()
    with pytest.raises(MeepoError):
        t.insert(T(np.arange(5000, dtype=np.int64), dev), torch.zeros(5000, 64, device=dev))  # > max_batch
    with pytest.raises(MeepoError):
        t.find(torch.zeros(4, dtype=torch.int32, device=dev))


def test_duplicates_last_wins(dev):
    dim, n = 64, 30000
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 500, size=n).astype(np.int64)          # ~60 occurrences per key
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    t = LookupTable(2048, dim, device=dev, max_batch=n); o = oracle.OracleTable(2048, dim)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    assert t.size() == o.size()
    q = np.arange(-5, 505, dtype=np.int64)
    out, found = t.find(T(q, dev)); eo, ef = o.find(q)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    rows2 = rng.standard_normal((n, dim)).astype(np.float32)
    keys2 = rng.integers(400, 700, size=n).astype(np.int64)
    fa = t.assign(T(keys2, dev), T(rows2, dev)); fo = o.assign(keys2, rows2)
    assert np.array_equal(fa.cpu().numpy(), fo)
    q = np.arange(350, 750, dtype=np.int64)
    out, found = t.find(T(q, dev)); eo, ef = o.find(q)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)


def test_reserved_keys_and_table_full(dev):
    t = LookupTable(32, 4, device=dev, max_batch=256); o = oracle.OracleTable(32, 4)
    keys = np.array([oracle.EMPTY_KEY, 5, oracle.RECLAIMED_KEY, 6], dtype=np.int64)
    ones = np.ones((4, 4), np.float32)
    t.insert(T(keys, dev), T(ones, dev)); o.insert(keys, ones)
    assert t.size() == o.size() == 2 and t.status() == o.status() == STATUS_RESERVED_KEY
    t2 = LookupTable(32, 4, device=dev, max_batch=256)
    pad = np.array([oracle.EMPTY_KEY, 9, oracle.EMPTY_KEY], dtype=np.int64)
    t2.insert(T(pad, dev), torch.ones(3, 4, device=dev)); t2.find_or_insert(T(pad, dev)); t2.remove(T(pad[:1], dev))
    assert t2.size() == 1 and t2.status() == 0, "EMPTY in a batch is padding: skipped silently"
    out, found = t.find(T(keys, dev)); eo, ef = o.find(keys)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    t.clear_status()
    many = synth.keys_np(4, 0, 100)
    t.insert(T(many, dev), torch.zeros(100, 4, device=dev))
    assert t.size() == 32 and t.status() & STATUS_TABLE_FULL
    _, found = t.find(T(many, dev))
    assert int(found.sum()) == 30      # which 30 is placement-order dependent (not observable), the count is not


def test_high_load_long_probes(dev):
    """Load factor 0.97: multi-bucket probe chains, every key still found, absent keys still absent."""
    dim, cap = 16, 16 * 256
    n = int(cap * 0.97)
    keys = synth.keys_np(13, 0, 2 * n); rows = synth.rows_np(keys[:n], dim, 1)
    t = LookupTable(cap, dim, device=dev, max_batch=2 * n); o = oracle.OracleTable(cap, dim)
    t.insert(T(keys[:n], dev), T(rows, dev)); o.insert(keys[:n], rows)
    out, found = t.find(T(keys, dev)); eo, ef = o.find(keys)
    assert t.size() == n and np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)


def test_remove_and_churn(dev):
    """remove + tombstone reuse vs the oracle: found-masks, survivors, re-insertion, a table kept ~full under churn."""
    dim, cap = 16, 16 * 300
    n = int(cap * 0.95)
    t = LookupTable(cap, dim, device=dev, max_batch=2 * n, optimizer=OPT_ADAGRAD, initial_accumulator=0.5)
    o = oracle.OracleTable(cap, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.5)
    keys = synth.keys_np(14, 0, n); rows = synth.rows_np(keys, dim, 1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    rng = np.random.default_rng(2)
    live = keys.copy()
    for rnd in range(6):
        victims = np.concatenate([rng.choice(live, size=n // 3, replace=False), synth.keys_np(500 + rnd, 0, 40), live[:5], live[:5]])
        fg = t.remove(T(victims, dev)); fo = o.remove(victims)
        assert np.array_equal(fg.cpu().numpy(), fo)
        live = np.setdiff1d(live, victims)
        assert t.size() == o.size() == live.size
        fresh = synth.keys_np(900 + rnd, 0, n - live.size)          # refill to the same fill level: must reuse tombstones
        fr = synth.rows_np(fresh, dim, 3 + rnd)
        t.insert(T(fresh, dev), T(fr, dev)); o.insert(fresh, fr)
        live = np.concatenate([live, fresh])
        assert t.size() == o.size() == live.size and t.status() == o.status() == 0
        q = np.concatenate([live[rng.permutation(live.size)[:2000]], victims[:500]])
        out, found = t.find(T(q, dev)); eo, ef = o.find(q)
        assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    g_ = [x.cpu().numpy() for x in t.export(with_state=True)[:3]]; o_ = o.export(with_state=True)[:3]
    a, b = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][a], o_[0][b]) and np.array_equal(g_[1][a], o_[1][b]) and np.array_equal(g_[2][a], o_[2][b])


def test_find_or_insert(dev):
    dim = 64
    kw = dict(initializer=INIT_UNIFORM, init_scale=0.05, init_seed=17, initial_accumulator=0.25)
    t = LookupTable(4096, dim, device=dev, max_batch=8192, optimizer=OPT_ADAGRAD, **kw)
    o = oracle.OracleTable(4096, dim, optimizer=oracle.OPT_ADAGRAD, **kw)
    rng = np.random.default_rng(3)
    for rnd in range(3):
        keys = rng.integers(0, 1500, size=3000).astype(np.int64) + 1000 * rnd
        out, found = t.find_or_insert(T(keys, dev)); eo, ef = o.find_or_insert(keys)
        assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
        assert t.size() == o.size()
    gk, gv, ga, _ = t.export(with_state=True); ok, ov, oa, _ = o.export(with_state=True)
    gk, gv, ga = sorted_export(gk.cpu().numpy(), gv.cpu().numpy(), ga.cpu().numpy()); ok, ov, oa = sorted_export(ok, ov, oa)
    assert np.array_equal(gk, ok) and np.array_equal(gv, ov) and np.array_equal(ga, oa)


def test_admission_policy(dev):
    """find_or_insert with an admission threshold (SPEC.md §3): an absent key is created only once the table's count-min sketch has seen
    it requested min_count times — this batch's occurrences included, decided after all of them were counted.  The expected behaviour is
    recomputed here: the sketch in numpy (same hashes, from oracle/pyspec.py's mix64), the table in the oracle."""
    from oracle import pyspec
    dim, cap, min_count = 16, 8192, 3
    A = (0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9)
    t = LookupTable(cap, dim, device=dev, max_batch=4096, admission=True, initializer=INIT_UNIFORM, init_scale=0.1, init_seed=5, default_value=-1.0)
    o = oracle.OracleTable(cap, dim, initializer=oracle.INIT_UNIFORM, init_scale=0.1, init_seed=5, default_value=-1.0)
    log2w = 12
    while (1 << log2w) < (t.capacity + 15) // 16:
        log2w += 1
    sketch = np.zeros((3, 1 << log2w), dtype=np.int64)
    idx = lambda k, r: pyspec.mix64((int(k) & (2**64 - 1)) ^ A[r]) >> (64 - log2w)   # noqa: E731
    universe = synth.keys_np(61, 0, 600)
    rng = np.random.default_rng(8)
    created_total = 0
    for step in range(12):
        bk = universe[np.minimum(rng.zipf(1.6, size=700) - 1, universe.size - 1)]
        bk[rng.integers(0, 700, 3)] = oracle.EMPTY_KEY
        out, found = t.find_or_insert(T(bk, dev), min_count=min_count)
        eo, ef = o.find(bk)                                  # present before the call
        absent = [k for k, f in zip(bk, ef) if not f and k > oracle.EMPTY_KEY + 1]
        for k in absent:
            for r in range(3):
                sketch[r, idx(k, r)] += 1
        admit = sorted({int(k) for k in absent if min(sketch[r, idx(k, r)] for r in range(3)) >= min_count})
        if admit:
            o.find_or_insert(np.array(admit, dtype=np.int64))
            created_total += len(admit)
        eo2, _ = o.find(bk)                                  # rows after the admitted keys were created; others: default row
        assert np.array_equal(found.cpu().numpy(), ef)
        assert np.array_equal(out.cpu().numpy(), eo2)
        assert t.size() == o.size()
        if step == 7:
            t.admission_decay(1)
            sketch >>= 1
    assert 0 < created_total < universe.size and t.status() == 0
    with pytest.raises(MeepoError):                          # a table without the sketch says so
        LookupTable(1024, dim, device=dev).find_or_insert(T(bk[:4], dev), min_count=2)


def test_dedup_sum(dev):
    dim, n = 64, 40000
    rng = np.random.default_rng(7)
    keys = np.concatenate([rng.integers(0, 3000, size=n - 5000), np.full(5000, 77)]).astype(np.int64)  # one heavy key
    keys[123] = oracle.EMPTY_KEY
    grads = rng.standard_normal((n, dim)).astype(np.float32)
    t = LookupTable(64, dim, device=dev, max_batch=n)
    uniq, gs, cnt, inv = t.dedup_sum(T(keys, dev), T(grads, dev), compact=True)
    uniq, gs, cnt, inv = uniq.cpu().numpy(), gs.cpu().numpy(), cnt.cpu().numpy(), inv.cpu().numpy()
    ou, ogs, oinv, ocnt = oracle.dedup_sum(keys, grads, dim)
    order, oorder = np.argsort(uniq), np.argsort(ou)
    assert np.array_equal(uniq[order], ou[oorder]) and np.array_equal(cnt[order], ocnt[oorder])
    np.testing.assert_allclose(gs[order], ogs[oorder], rtol=RTOL, atol=ATOL)
    assert inv[123] == -1 and np.array_equal(uniq[inv[inv >= 0]], keys[inv >= 0])
    # the scratch is left clean: a second, different batch gives a correct answer too
    u2, _, c2, _ = t.dedup_sum(T(keys[:100], dev), compact=True)
    assert int(c2.sum()) == 100 - (1 if 123 < 100 else 0) and len(np.unique(u2.cpu().numpy())) == u2.numel()


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
@pytest.mark.parametrize("dim", [16, 64, 128])
def test_optimizer_parity(dev, opt, dim):
    n_keys, steps, batch = 20000, 4, 30000
    keys = synth.keys_np(31, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    t = LookupTable(32768, dim, device=dev, optimizer=kind, max_batch=batch, initial_accumulator=0.1)
    o = oracle.OracleTable(32768, dim, optimizer=okind, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    rng = np.random.default_rng(41)
    for s in range(steps):
        # zipf-ish skew: many duplicates incl. a few heavy keys, plus absent keys that must be ignored
        idx = np.minimum(rng.zipf(1.3, size=batch) - 1, n_keys - 1)
        bk = keys[idx]
        bk[rng.integers(0, batch, 50)] = synth.keys_np(77, s * 50, 50)
        g = (rng.standard_normal((batch, dim)) * 0.01).astype(np.float32)
        if opt == "adagrad":
            t.apply_adagrad(T(bk, dev), T(g, dev), lr=0.01, eps=1e-10); o.apply_adagrad(bk, g, 0.01, 1e-10)
        else:
            t.apply_adam(T(bk, dev), T(g, dev), lr=0.001, step=s + 1); o.apply_adam(bk, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
    assert t.size() == o.size() == n_keys
    g_ = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
    o_ = o.export(with_state=True)
    order, oorder = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][order], o_[0][oorder])
    for a, b in zip(g_[1:], o_[1:]):
        if b is not None:
            np.testing.assert_allclose(a[order], b[oorder], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_optimizer_full_batch_of_medium_groups(dev, opt):
    """n == max_batch and EVERY key repeated 33..40 times: each group is 'big' (> 32 occurrences) and needs TWO fp64
    partial-sum rows, the worst case for the partial-sum block (about n/32 + n/33 rows; a block sized for n/32 + 1 rows
    was overrun by exactly this shape)."""
    dim, batch = 64, 1 << 16
    rng = np.random.default_rng(77)
    reps = rng.integers(33, 41, size=batch // 33 + 1)
    reps = reps[np.cumsum(reps) <= batch]
    n_keys = reps.size
    keys = synth.keys_np(57, 0, n_keys + 1); rows = synth.rows_np(keys, dim, 2)
    bk = np.repeat(keys[:n_keys], reps)
    bk = np.concatenate([bk, np.full(batch - bk.size, keys[n_keys])])   # the remainder: one more (small or big) group
    rng.shuffle(bk)
    assert bk.size == batch
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    t = LookupTable(8192, dim, device=dev, optimizer=kind, max_batch=batch, initial_accumulator=0.1)
    o = oracle.OracleTable(8192, dim, optimizer=okind, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    # a canary table allocated right behind the first one's scratch: an overrun of the partial-sum block would land in it
    canary = LookupTable(8192, dim, device=dev, optimizer=kind, max_batch=4096)
    canary.insert(T(keys, dev), T(rows, dev))
    for s in range(2):
        g = (rng.standard_normal((batch, dim)) * 0.01).astype(np.float32)
        if opt == "adagrad":
            t.apply_adagrad(T(bk, dev), T(g, dev), lr=0.01, eps=1e-10); o.apply_adagrad(bk, g, 0.01, 1e-10)
        else:
            t.apply_adam(T(bk, dev), T(g, dev), lr=0.001, step=s + 1); o.apply_adam(bk, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
    assert t.status() == 0
    g_ = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
    o_ = o.export(with_state=True)
    order, oorder = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][order], o_[0][oorder])
    for a, b in zip(g_[1:], o_[1:]):
        if b is not None:
            np.testing.assert_allclose(a[order], b[oorder], rtol=RTOL, atol=ATOL)
    got, found = canary.find(T(keys, dev))
    assert bool(found.all()) and np.array_equal(got.cpu().numpy(), rows)


@pytest.mark.parametrize("opt,dim,layout", [("adagrad", 64, "clustered"), ("adagrad", 64, "spread"), ("adam", 128, "mixed"),
                                            ("adagrad", 16, "mixed"), ("adam", 24, "spread")])
def test_optimizer_every_group_size(dev, opt, dim, layout):
    """One batch holds a key of EVERY multiplicity 1..44 plus 64, 65, 100, 333 and 2100, so that each way a duplicate group can be
    finished is taken and its edges are crossed: the inline list of a group-table entry (the claiming block's occurrences 1..8, eight
    of other blocks), the filed groups of one chunk (up to 32), the groups with fp64 partial-sum rows (33 and more).  'clustered'
    keeps a key's occurrences adjacent (one block of the grouping kernel sees them all), 'spread' puts them 1031 positions apart
    (every occurrence in another block), 'mixed' does both at random.  Plain, located and indexed applies must all match the oracle."""
    rng = np.random.default_rng(7 + dim)
    sizes = list(range(1, 45)) + [64, 65, 100, 333, 2100]
    n_keys = len(sizes) * 3
    keys = synth.keys_np(123, 0, n_keys + 500); rows = synth.rows_np(keys, dim, 2)
    reps = np.array(sizes * 3)
    bk = np.repeat(keys[:n_keys], reps)
    filler = keys[n_keys:n_keys + 400]                       # single keys between the groups
    n = 1031 * ((bk.size + filler.size) // 1031 + 1)
    batch = np.full(n, oracle.EMPTY_KEY, dtype=np.int64)    # padding where nothing lands
    if layout == "clustered":
        order = np.arange(bk.size)
    elif layout == "spread":
        k = np.arange(bk.size)
        order = (k % (n // 1031)) * 1031 + k // (n // 1031)  # neighbours in bk land 1031 positions apart
        assert np.unique(order).size == bk.size
    else:
        order = rng.permutation(n)[:bk.size]
        half = rng.random(bk.size) < 0.5                     # half of the occurrences stay next to their neighbours
        order[half] = np.sort(order[half])
    batch[order] = bk
    free = np.flatnonzero(batch == oracle.EMPTY_KEY)
    batch[free[:filler.size]] = filler
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    mk = lambda: LookupTable(4096, dim, device=dev, optimizer=kind, max_batch=n, initial_accumulator=0.1)
    ta, tb, tc = mk(), mk(), mk()
    o = oracle.OracleTable(4096, dim, optimizer=okind, initial_accumulator=0.1)
    for t in (ta, tb, tc):
        t.insert(T(keys[:n_keys + 400], dev), T(rows[:n_keys + 400], dev))
    o.insert(keys[:n_keys + 400], rows[:n_keys + 400])
    bkt = T(batch, dev)
    for s in range(2):
        pool = (rng.standard_normal((n // 3 + 1, dim)) * 0.02).astype(np.float32)     # indexed apply: three positions share a grad row
        gi = rng.integers(0, pool.shape[0], n).astype(np.int64)
        g = pool[gi]
        _, _, slots = tb.find_located(bkt)
        if opt == "adagrad":
            ta.apply_adagrad(bkt, T(g, dev), lr=0.05)
            tb.apply_adagrad(bkt, T(g, dev), lr=0.05, slots=slots)
            tc.apply_adagrad(bkt, T(pool, dev), lr=0.05, grad_index=T(gi, dev))
            o.apply_adagrad(batch, g, 0.05, 1e-10)
        else:
            ta.apply_adam(bkt, T(g, dev), lr=0.01, step=s + 1)
            tb.apply_adam(bkt, T(g, dev), lr=0.01, step=s + 1, slots=slots)
            tc.apply_adam(bkt, T(pool, dev), lr=0.01, step=s + 1, grad_index=T(gi, dev))
            o.apply_adam(batch, g, 0.01, 0.9, 0.999, 1e-8, s + 1)
    eo = o.export(with_state=True)
    io = np.argsort(eo[0])
    for t in (ta, tb, tc):
        assert t.status() == 0
        e = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
        it = np.argsort(e[0])
        assert np.array_equal(e[0][it], eo[0][io])
        for x, z in zip(e[1:], eo[1:]):
            if z is not None:
                np.testing.assert_allclose(x[it], z[io], rtol=RTOL, atol=ATOL)
    # the scratch is left clean: a batch of distinct keys right behind it
    g1 = (rng.standard_normal((400, dim)) * 0.02).astype(np.float32)
    if opt == "adagrad":
        ta.apply_adagrad(T(filler, dev), T(g1, dev), lr=0.05); o.apply_adagrad(filler, g1, 0.05, 1e-10)
    else:
        ta.apply_adam(T(filler, dev), T(g1, dev), lr=0.01, step=3); o.apply_adam(filler, g1, 0.01, 0.9, 0.999, 1e-8, 3)
    got, found = ta.find(T(filler, dev))
    exp, _ = o.find(filler)
    assert bool(found.all())
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=RTOL, atol=ATOL)


def _keys_of_apply_bucket_zero(count, seed):
    """Distinct keys whose mix64 has 13 leading zero bits: for ANY bucket count up to 8192 the bucketed apply puts them all into bucket 0
    (and the table into the first 1/8192 of its buckets)."""
    rng = np.random.default_rng(seed)
    got = []
    while sum(len(g) for g in got) < count:
        cand = rng.integers(-(1 << 62), 1 << 62, size=1 << 22, dtype=np.int64)
        mix, _, _ = oracle.hash_batch(cand, 1, 1)
        got.append(cand[mix < (np.uint64(1) << np.uint64(51))])
    return np.unique(np.concatenate(got))[:count]


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
@pytest.mark.parametrize("kernel", ["auto", "lean", "full"])
@pytest.mark.parametrize("case", ["one_key", "one_bucket_many_keys", "forty_hot_keys", "one_bucket_two_keys", "bucket_of_900_distinct", "bucket_of_300_warm"])
def test_bucketed_apply_extremes(dev, case, opt, kernel):
    """The rare ways through the bucketed apply (meepo_apply.hip), each forced by construction, plain and located, against the oracle:
    one_key — a single key fills 400K of a 410K-position batch: ~780 slabs of one bucket each emit a record of that key, more records of ONE key
    than a merge pass holds (mono_pass);  one_bucket_many_keys — 3000 keys that all fall into apply bucket 0, 100+ occurrences each: every slab
    emits hundreds of records, the merge has far more records than one pass holds and splits them by hash prefix (the DFS stack);
    forty_hot_keys — 40 keys of ~6000 occurrences in a uniform batch: forty split buckets merge side by side, spare blocks loop over slabs;
    one_bucket_two_keys — two keys of one bucket, 150K occurrences each: the prefix split must separate exactly two keys;
    bucket_of_900_distinct — 900 keys of apply bucket 0, once each, in a small batch: ONE block takes a bucket of ~1000 positions whole (two positions
    per thread) with its LDS hash table filled almost to the last slot;  bucket_of_300_warm — 300 keys of bucket 0 with 1..6 occurrences each: the
    same path with runs.
    kernel: which apply kernel takes the batches — "lean" (block = bucket; a split bucket is taken by its own block one key at a time: what the
    FIRST skewed batch of a stream gets), "full" (slabs, pending records, merges; from the second step on also the hot keys' own buckets, which the
    first step's kernel reported), "auto" (the library's choice: lean for step 0, full for step 1)."""
    dim, n_bg = 64, 20000
    rng = np.random.default_rng(5)
    bg = synth.keys_np(321, 0, n_bg)
    if case == "one_key":
        hot = synth.keys_np(322, 0, 1); reps = np.array([400_000]); n_fill = 10_000
    elif case == "one_bucket_many_keys":
        hot = _keys_of_apply_bucket_zero(3000, 7); reps = rng.integers(100, 140, size=3000); n_fill = 20_000
    elif case == "forty_hot_keys":
        hot = synth.keys_np(323, 0, 40); reps = rng.integers(5000, 7000, size=40); n_fill = 150_000
    elif case == "one_bucket_two_keys":
        hot = _keys_of_apply_bucket_zero(2, 9); reps = np.array([150_000, 150_001]); n_fill = 5_000
    elif case == "bucket_of_900_distinct":
        hot = _keys_of_apply_bucket_zero(900, 11); reps = np.ones(900, dtype=np.int64); n_fill = 5_000
    else:
        hot = _keys_of_apply_bucket_zero(300, 12); reps = rng.integers(1, 7, size=300); n_fill = 3_000
    keys = np.unique(np.concatenate([bg, hot]))
    rows = synth.rows_np(keys, dim, 2)
    bk = np.concatenate([np.repeat(hot, reps), bg[rng.integers(0, n_bg, n_fill)], synth.keys_np(324, 0, 50)])   # + 50 absent keys
    rng.shuffle(bk)
    n = bk.size
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    mk = lambda: LookupTable(1 << 17, dim, device=dev, optimizer=kind, max_batch=max(n, keys.size), initial_accumulator=0.1)
    ta, tb = mk(), mk()
    o = oracle.OracleTable(1 << 17, dim, optimizer=okind, initial_accumulator=0.1)
    for t in (ta, tb):
        t.insert(T(keys, dev), T(rows, dev))
        t.set_tuning("apply_kernel", {"auto": -1, "lean": 0, "full": 1}[kernel])
    o.insert(keys, rows)
    bkt = T(bk, dev)
    for s in range(3 if kernel == "full" else 2):
        torch.cuda.synchronize()   # (the host sizes step s + 1 by what step s reported: hot keys' buckets exist from the second full step on)
        g = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
        gt = T(g, dev)
        _, _, slots = tb.find_located(bkt, prepare_apply=(s == 1))
        if opt == "adagrad":
            ta.apply_adagrad(bkt, gt, lr=0.05); tb.apply_adagrad(bkt, gt, lr=0.05, slots=slots); o.apply_adagrad(bk, g, 0.05, 1e-10)
        else:
            ta.apply_adam(bkt, gt, lr=0.01, step=s + 1); tb.apply_adam(bkt, gt, lr=0.01, step=s + 1, slots=slots)
            o.apply_adam(bk, g, 0.01, 0.9, 0.999, 1e-8, s + 1)
    eo = o.export(with_state=True)
    io = np.argsort(eo[0])
    for t in (ta, tb):
        assert t.status() == 0
        e = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
        it = np.argsort(e[0])
        assert np.array_equal(e[0][it], eo[0][io])
        for x, z in zip(e[1:], eo[1:]):
            if z is not None:
                np.testing.assert_allclose(x[it], z[io], rtol=RTOL, atol=ATOL)
    # the scratch is left clean: a batch of distinct keys right behind it, bit-exact
    g1 = (rng.standard_normal((n_bg, dim)) * 0.02).astype(np.float32)
    if opt == "adagrad":
        ta.apply_adagrad(T(bg, dev), T(g1, dev), lr=0.05); o.apply_adagrad(bg, g1, 0.05, 1e-10)
    else:
        ta.apply_adam(T(bg, dev), T(g1, dev), lr=0.01, step=3); o.apply_adam(bg, g1, 0.01, 0.9, 0.999, 1e-8, 3)
    got, found = ta.find(T(bg, dev))
    exp, _ = o.find(bg)
    assert bool(found.all()) and ta.status() == 0
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("sync_every_step", [True, False], ids=["host_in_step", "host_runs_ahead"])
def test_hot_key_set_churn_across_operators(dev, sync_every_step):
    """The hot-key set is state that one operator's kernel leaves for the NEXT partition of the same table, whichever operator that is: ten steps that
    alternate plain applies, located applies behind the training forward, dedup_keys and assign, each on a batch of 300K positions with ~200 hot keys
    (more than the 128 the set holds) drawn from a pool that rotates from step to step (keys enter and leave the set; the set's two copies alternate with the
    batch parity), with and without a host synchronisation between the steps (with: the next partition is sized by what the last kernel reported; without:
    by whatever stale report the host word holds).  Every step's results against the oracle."""
    dim, n_bg, n_pool = 16, 60_000, 400
    rng = np.random.default_rng(77)
    bg = synth.keys_np(611, 0, n_bg)
    pool = synth.keys_np(612, 0, n_pool)
    keys = np.concatenate([bg, pool]); rows = synth.rows_np(keys, dim, 2)
    t = LookupTable(1 << 18, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=400_000, initial_accumulator=0.1)
    o = oracle.OracleTable(1 << 18, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    ops = ["apply", "located", "dedup", "assign", "apply", "dedup", "located", "assign", "apply", "located"]
    for s, op in enumerate(ops):
        hot = pool[(np.arange(200) + 57 * s) % n_pool]                      # the pool rotates: a third of the hot keys are new each step
        reps = rng.integers(280, 900, size=200); reps[:3] = (20_000, 9_000, 1_100)   # a few VERY hot ones: split buckets even in their own bucket
        bk = np.concatenate([np.repeat(hot, reps), bg[rng.integers(0, n_bg, 60_000)], synth.keys_np(613 + s, 0, 40)])   # + 40 absent keys
        bk = bk[:300_000] if bk.size > 300_000 else bk
        rng.shuffle(bk)
        bk[5] = oracle.EMPTY_KEY
        n = bk.size
        bkt = T(bk, dev)
        if op in ("apply", "located"):
            g = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
            if op == "apply":
                t.apply_adagrad(bkt, T(g, dev), lr=0.05)
            else:
                _, _, slots = t.find_located(bkt, prepare_apply=True)
                t.apply_adagrad(bkt, T(g, dev), lr=0.05, slots=slots)
            o.apply_adagrad(bk, g, 0.05, 1e-10)
        elif op == "dedup":
            uniq, inverse = t.dedup_keys(bkt, miss_index=-7)
            uniq, inverse = uniq.cpu().numpy(), inverse.cpu().numpy()
            ou = oracle.dedup_sum(bk, None, dim)[0]
            at = np.flatnonzero(uniq != oracle.EMPTY_KEY)
            assert at.size == ou.size and np.array_equal(np.sort(uniq[at]), np.sort(ou)), f"step {s}: distinct keys"
            valid = bk != oracle.EMPTY_KEY
            assert np.array_equal(uniq[inverse[valid]], bk[valid]) and (inverse[~valid] == -7).all(), f"step {s}: inverse"
        else:
            v = rng.standard_normal((n, dim)).astype(np.float32)
            f = t.assign(bkt, T(v, dev)).cpu().numpy()
            fo = o.assign(bk, v)
            assert np.array_equal(f.astype(bool), np.asarray(fo).astype(bool)), f"step {s}: assign found mask"
        if sync_every_step:
            torch.cuda.synchronize()
    assert t.status() == 0
    e = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
    eo = o.export(with_state=True)
    it, io = np.argsort(e[0]), np.argsort(eo[0])
    assert np.array_equal(e[0][it], eo[0][io])
    for x, z in zip(e[1:], eo[1:]):
        if z is not None:
            np.testing.assert_allclose(x[it], z[io], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("n", [2_900_000, 9_500_000], ids=["2.9M", "9.5M"])
def test_apply_beyond_the_bucketed_limit(dev, n):
    """Batches beyond what round 3's bucketed apply took (8192 buckets x 352 positions = 2.7M keys; the group-table apply behind it is gone):
    2.9M keys = buckets of ~380 positions, still one block each; 9.5M keys = every bucket holds more than a block takes whole (1024) and goes
    through its slabs, pending records and merge passes.  Also behind the training forward (mee_find_located_prepare); same results as the
    oracle, and the next small batch finds the scratch clean."""
    dim, n_keys = 4, 400_000
    keys = synth.keys_np(171, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    rng = np.random.default_rng(8)
    bk = keys[rng.integers(0, n_keys, n)]           # ~7 occurrences per key
    bk[:50_000] = keys[3]                            # and one hot key
    bk[rng.integers(0, n, 100)] = synth.keys_np(172, 0, 100)   # absent
    t = LookupTable(1 << 20, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=n + 100_000, initial_accumulator=0.1)
    o = oracle.OracleTable(1 << 20, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    bkt = T(bk, dev)
    for s in range(2):
        g = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
        if s == 0:
            t.apply_adagrad(bkt, T(g, dev), lr=0.05)
        else:
            _, _, slots = t.find_located(bkt, prepare_apply=True)
            t.apply_adagrad(bkt, T(g, dev), lr=0.05, slots=slots)
        o.apply_adagrad(bk, g, 0.05, 1e-10)
    g1 = (rng.standard_normal((5000, dim)) * 0.01).astype(np.float32)
    t.apply_adagrad(T(keys[:5000], dev), T(g1, dev), lr=0.05); o.apply_adagrad(keys[:5000], g1, 0.05, 1e-10)
    assert t.status() == 0
    e = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
    eo = o.export(with_state=True)
    it, io = np.argsort(e[0]), np.argsort(eo[0])
    assert np.array_equal(e[0][it], eo[0][io])
    for x, z in zip(e[1:], eo[1:]):
        if z is not None:
            np.testing.assert_allclose(x[it], z[io], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("dim", [64, 24])
def test_find_or_insert_located(dev, dim):
    """find_or_insert_located == find_or_insert (rows, found, table contents) and its handles are the slots mee_locate reports
    afterwards — for present keys, new keys, duplicates of new keys, padding and a full table (-1)."""
    n_keys = 3000
    keys = synth.keys_np(91, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    mk = lambda cap: LookupTable(cap, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=8192, initializer=INIT_UNIFORM, init_scale=0.05, init_seed=5)
    ta, tb = mk(8192), mk(8192)
    for t in (ta, tb):
        t.insert(T(keys[:2000], dev), T(rows[:2000], dev))
    rng = np.random.default_rng(4)
    batch = np.concatenate([keys[rng.integers(0, n_keys, 5000)], keys[2500:2600], keys[2500:2600]])   # present, new, new twice
    batch[rng.integers(0, batch.size, 5)] = oracle.EMPTY_KEY
    rng.shuffle(batch)
    oa, fa = ta.find_or_insert(T(batch, dev))
    ob, fb, slots = tb.find_or_insert_located(T(batch, dev))
    assert torch.equal(oa, ob) and torch.equal(fa, fb)
    loc, lf = tb.locate(T(batch, dev))
    # a handle = the slot mee_locate reports (bits 0..39) + the table's layout epoch (bits 40..61)
    assert torch.equal(torch.where(slots >= 0, slots & _lib.HANDLE_SLOT_MASK, slots), torch.where(lf.bool(), loc, torch.full_like(loc, -1)))
    assert bool(((slots >= 0) == T(batch != oracle.EMPTY_KEY, dev)).all())
    ea, eb = ta.export(with_state=True), tb.export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and torch.equal(ea[1][ia], eb[1][ib]) and torch.equal(ea[2][ia], eb[2][ib])
    # the handles drive the apply of the same step
    g = (rng.standard_normal((batch.size, dim)) * 0.01).astype(np.float32)
    ta.apply_adagrad(T(batch, dev), T(g, dev), lr=0.01); tb.apply_adagrad(T(batch, dev), T(g, dev), lr=0.01, slots=slots)
    ea, eb = ta.export(with_state=True), tb.export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    np.testing.assert_allclose(ea[1][ia].cpu().numpy(), eb[1][ib].cpu().numpy(), rtol=RTOL, atol=ATOL)
    # the training forward of a growing vocabulary: the same call with the backward's partition in its first launch
    batch2 = np.concatenate([keys[rng.integers(0, n_keys, 4000)], synth.keys_np(93, 0, 300), synth.keys_np(93, 0, 300)])   # present, new, new twice
    rng.shuffle(batch2)
    b2 = T(batch2, dev)
    oa, fa = ta.find_or_insert(b2)
    ob, fb, slots = tb.find_or_insert_located(b2, prepare_apply=True)
    assert torch.equal(oa, ob) and torch.equal(fa, fb) and bool((slots >= 0).all())
    # a mutator between the training forward and its backward (an eviction hook, a second lookup that creates keys, growth): it drops the
    # partition the forward left and the apply partitions its batch again — same result (here: rows rewritten with what they hold already)
    tb.insert(b2[:4], ob[:4])
    g = (rng.standard_normal((batch2.size, dim)) * 0.01).astype(np.float32)
    ta.apply_adagrad(b2, T(g, dev), lr=0.01); tb.apply_adagrad(b2, T(g, dev), lr=0.01, slots=slots)
    ea, eb = ta.export(with_state=True), tb.export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and ta.status() == tb.status() == 0
    np.testing.assert_allclose(ea[1][ia].cpu().numpy(), eb[1][ib].cpu().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ea[2][ia].cpu().numpy(), eb[2][ib].cpu().numpy(), rtol=RTOL, atol=ATOL)
    # a full table: keys that cannot be created report -1 and the default row
    small = LookupTable(64, dim, device=dev, max_batch=8192)
    o, f, s_ = small.find_or_insert_located(T(keys[:1000], dev))
    assert small.status() & STATUS_TABLE_FULL and int((s_ >= 0).sum()) == small.size() and int((s_ < 0).sum()) == 1000 - small.size()


@pytest.mark.parametrize("opt,dim", [("adagrad", 64), ("adam", 128), ("adagrad", 24)])
def test_located_apply_equals_plain_apply(dev, opt, dim):
    """find_located + apply_*(slots=…) — the forward's slot handles instead of a probe — must give the table the plain apply
    gives (and the oracle's), with duplicates, absent keys (handle -1), reserved keys and both settings of the side-stream knob."""
    n_keys, batch = 30000, 20000
    keys = synth.keys_np(83, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    mk = lambda: LookupTable(65536, dim, device=dev, optimizer=kind, max_batch=n_keys, initial_accumulator=0.1)
    ta, tb = mk(), mk()
    o = oracle.OracleTable(65536, dim, optimizer=okind, initial_accumulator=0.1)
    for t in (ta, tb):
        t.insert(T(keys, dev), T(rows, dev))
    o.insert(keys, rows)
    rng = np.random.default_rng(3)
    for s in range(4):
        idx = np.minimum(rng.zipf(1.4, size=batch) - 1, n_keys - 1) if s % 2 else rng.integers(0, n_keys, batch)
        bk = keys[idx]
        bk[rng.integers(0, batch, 40)] = synth.keys_np(91, s * 40, 40)      # absent keys
        bk[rng.integers(0, batch, 3)] = oracle.EMPTY_KEY                     # padding
        g = (rng.standard_normal((batch, dim)) * 0.01).astype(np.float32)
        bkt = T(bk, dev)
        out, found, slots = tb.find_located(bkt)
        eo, ef = ta.find(T(bk, dev))
        assert torch.equal(out, eo) and torch.equal(found, ef)
        assert torch.equal(slots >= 0, found.bool())
        if opt == "adagrad":
            ta.apply_adagrad(T(bk, dev), T(g, dev), lr=0.01); tb.apply_adagrad(bkt, T(g, dev), lr=0.01, slots=slots)
            o.apply_adagrad(bk, g, 0.01, 1e-10)
        else:
            ta.apply_adam(T(bk, dev), T(g, dev), lr=0.001, step=s + 1); tb.apply_adam(bkt, T(g, dev), lr=0.001, step=s + 1, slots=slots)
            o.apply_adam(bk, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
    ea, eb = ta.export(with_state=True), tb.export(with_state=True)
    oe = o.export(with_state=True)
    ia, ib, io = torch.argsort(ea[0]), torch.argsort(eb[0]), np.argsort(oe[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and np.array_equal(ea[0][ia].cpu().numpy(), oe[0][io])
    for xa, xb, xo in zip(ea[1:], eb[1:], oe[1:]):
        if xo is not None:
            torch.testing.assert_close(xa[ia], xb[ib], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(xb[ib].cpu().numpy(), xo[io], rtol=RTOL, atol=ATOL)
    # out-of-range / negative handles are ignored, never dereferenced
    bad = torch.full((8,), 1 << 60, dtype=torch.int64, device=dev); bad[::2] = -1
    kk = T(synth.keys_np(95, 0, 8), dev)   # absent keys, so that nothing should change
    if opt == "adagrad":
        tb.apply_adagrad(kk, torch.ones(8, dim, device=dev), lr=0.1, slots=bad)
    else:
        tb.apply_adam(kk, torch.ones(8, dim, device=dev), lr=0.1, step=9, slots=bad)
    e2 = tb.export(with_state=True)
    i2 = torch.argsort(e2[0])
    assert torch.equal(e2[1][i2], eb[1][ib]) and tb.status() == _lib.STATUS_STALE_HANDLE   # a handle of no epoch of this table: flagged, not followed
    tb.clear_status()
    # handles kept across a remove are STALE: the freed slot may hold another key by now, so the located apply must not touch it
    kq = T(keys[:64], dev)
    _, _, h_old = tb.find_located(kq)
    tb.remove(T(keys[:8], dev))                                   # frees 8 slots; every handle made before is of an earlier layout epoch
    tb.insert(T(synth.keys_np(97, 0, 4000), dev), torch.zeros(4000, dim, device=dev))   # some of the freed slots are taken by other keys
    before = tb.export(with_state=True)
    g1 = torch.ones(64, dim, device=dev)
    if opt == "adagrad":
        tb.apply_adagrad(kq, g1, lr=0.1, slots=h_old)
    else:
        tb.apply_adam(kq, g1, lr=0.1, step=11, slots=h_old)
    after = tb.export(with_state=True)
    ib_, ia_ = torch.argsort(before[0]), torch.argsort(after[0])
    assert torch.equal(before[1][ib_], after[1][ia_]) and torch.equal(before[2][ib_], after[2][ia_]), "a stale handle updated a row"
    assert tb.status() & _lib.STATUS_STALE_HANDLE
    tb.clear_status()
    _, _, h_new = tb.find_located(kq)                             # fresh handles work again (keys[:8] are absent: -1)
    assert bool((h_new[:8] == -1).all()) and bool((h_new[8:] >= 0).all())
    if opt == "adagrad":
        tb.apply_adagrad(kq, g1, lr=0.1, slots=h_new)
    else:
        tb.apply_adam(kq, g1, lr=0.1, step=12, slots=h_new)
    assert tb.status() == 0 and not torch.equal(tb.find(kq[8:])[0], ta.find(kq[8:])[0])


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
@pytest.mark.parametrize("dim,n_keys,batch", [(64, 200000, 131072), (128, 3000, 2000), (24, 50000, 40000)])
def test_training_forward_prepares_the_apply(dev, opt, dim, n_keys, batch):
    """find_located(prepare_apply=True) — ONE launch: the located find + the partition of the step's apply, run by the launch's first blocks —
    followed by apply_*(slots=…) must leave the table exactly as find_located + apply_* does, and return the same rows; uniform and skewed
    batches (hot keys: split buckets), absent keys, padding; a prepared apply can also be discarded."""
    keys = synth.keys_np(61, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    kind = OPT_ADAGRAD if opt == "adagrad" else OPT_ADAM
    mk = lambda: LookupTable(int(n_keys * 1.5), dim, device=dev, optimizer=kind, max_batch=batch, initial_accumulator=0.1)
    ta, tb = mk(), mk()
    for t in (ta, tb):
        for s_ in range(0, n_keys, batch):
            t.insert(T(keys[s_:s_ + batch], dev), T(rows[s_:s_ + batch], dev))
    rng = np.random.default_rng(9)
    for step in range(4):
        idx = np.minimum(rng.zipf(1.2, size=batch) - 1, n_keys - 1) if step % 2 else rng.integers(0, n_keys, batch)
        bk_ = keys[idx]
        bk_[rng.integers(0, batch, 30)] = synth.keys_np(62, step * 30, 30)     # absent
        bk_[rng.integers(0, batch, 3)] = oracle.EMPTY_KEY                       # padding
        kt = T(bk_, dev)
        g = torch.randn(batch, dim, device=dev) * 0.01
        oa, fa, sa = ta.find_located(kt)
        ob, fb, sb = tb.find_located(kt, prepare_apply=True)
        assert torch.equal(oa, ob) and torch.equal(fa, fb) and torch.equal(sa >= 0, sb >= 0)   # (two tables: their slots differ)
        if step == 2:   # a prepared apply may be dropped (and made again)
            tb.apply_discard()
            tb.find_located(kt, out=ob, found=fb, slots=sb, prepare_apply=True)
        if opt == "adagrad":
            ta.apply_adagrad(kt, g, lr=0.01, slots=sa); tb.apply_adagrad(kt, g, lr=0.01, slots=sb)
        else:
            ta.apply_adam(kt, g, lr=0.001, step=step + 1, slots=sa); tb.apply_adam(kt, g, lr=0.001, step=step + 1, slots=sb)
    ea, eb = ta.export(with_state=True), tb.export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and ta.status() == tb.status() == 0
    for xa, xb in zip(ea[1:], eb[1:]):
        if xa is not None:
            torch.testing.assert_close(xa[ia], xb[ib], rtol=RTOL, atol=ATOL)


def test_optimizer_unique_keys_bit_exact(dev):
    """No duplicates -> no reduction-order freedom: the HIP update must equal the oracle bit for bit."""
    dim, n = 64, 10000
    keys = synth.keys_np(5, 0, n); rows = synth.rows_np(keys, dim, 2)
    g = (synth.rows_np(keys, dim, 6) * 0.02).astype(np.float32)
    for kind, okind in ((OPT_ADAGRAD, oracle.OPT_ADAGRAD), (OPT_ADAM, oracle.OPT_ADAM)):
        t = LookupTable(16384, dim, device=dev, optimizer=kind, max_batch=n); o = oracle.OracleTable(16384, dim, optimizer=okind)
        t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
        for s in range(3):
            if kind == OPT_ADAGRAD:
                t.apply_adagrad(T(keys, dev), T(g, dev), lr=0.01); o.apply_adagrad(keys, g, 0.01, 1e-10)
            else:
                t.apply_adam(T(keys, dev), T(g, dev), lr=0.001, step=s + 1); o.apply_adam(keys, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
        out, _ = t.find(T(keys, dev)); eo, _ = o.find(keys)
        assert np.array_equal(out.cpu().numpy(), eo)


def test_optimizer_vs_torch_golden_on_gpu(dev):
    z = np.load(os.path.join(GOLDEN, "optimizer_golden.npz"))
    for dim in (16, 64):
        w0, idx, grads = z[f"w0_{dim}"], z[f"idx_{dim}"], z[f"grads_{dim}"]
        keys = synth.keys_np(21, 0, w0.shape[0])
        for name in ("adagrad", "adam"):
            t = LookupTable(1024, dim, device=dev, optimizer=OPT_ADAGRAD if name == "adagrad" else OPT_ADAM, max_batch=1024,
                            initial_accumulator=0.1 if name == "adagrad" else 0.0)
            t.insert(T(keys, dev), T(w0, dev))
            for s in range(idx.shape[0]):
                if name == "adagrad":
                    t.apply_adagrad(T(keys[idx[s]], dev), T(grads[s], dev), lr=0.05, eps=1e-10)
                else:
                    t.apply_adam(T(keys[idx[s]], dev), T(grads[s], dev), lr=0.01, step=s + 1)
            got, _ = t.find(T(keys, dev))
            np.testing.assert_allclose(got.cpu().numpy(), z[f"{name}_w_{dim}"], rtol=2e-6, atol=1e-7)


def test_optimizer_vs_torch_golden_more_cases_on_gpu(dev):
    """Round 4's golden cases (dim 128, non-default eps / betas, initial accumulator 0 and 0.1) on the device, plain and located applies."""
    z = np.load(os.path.join(GOLDEN, "optimizer_golden.npz"))
    cases = json.loads(str(z["cases_json"]))
    for c in cases:
        t_ = c["tag"]
        w0, idx, grads = z[f"c_{t_}_w0"], z[f"c_{t_}_idx"], z[f"c_{t_}_grads"]
        keys = synth.keys_np(22, 0, w0.shape[0])
        adagrad = c["opt"] == "adagrad"
        for located in (False, True):
            t = LookupTable(1024, c["dim"], device=dev, optimizer=OPT_ADAGRAD if adagrad else OPT_ADAM, max_batch=1024, initial_accumulator=c.get("acc0", 0.0))
            t.insert(T(keys, dev), T(w0, dev))
            for s in range(idx.shape[0]):
                kt = T(keys[idx[s]], dev)
                slots = t.find_located(kt)[2] if located else None
                if adagrad:
                    t.apply_adagrad(kt, T(grads[s], dev), lr=c["lr"], eps=c["eps"], slots=slots)
                else:
                    t.apply_adam(kt, T(grads[s], dev), lr=c["lr"], beta1=c["beta1"], beta2=c["beta2"], eps=c["eps"], step=s + 1, slots=slots)
            got, _ = t.find(T(keys, dev))
            np.testing.assert_allclose(got.cpu().numpy(), z[f"c_{t_}_w"], rtol=2e-6, atol=1e-7, err_msg=f"{t_} located={located}")


@pytest.mark.parametrize("g", [1, 2, 3, 8])
def test_partition_and_permute(dev, g):
    n = 100_003
    keys = synth.keys_np(8, 0, n)
    r = Router(g, n, device=dev)
    send, counts, perm = r.partition(T(keys, dev))
    os_, oc, op = oracle.partition(keys, g)
    assert np.array_equal(send.cpu().numpy(), os_) and np.array_equal(counts.cpu().numpy(), oc) and np.array_equal(perm.cpu().numpy(), op)
    rows = synth.rows_np(keys, 16, 3)
    fwd = r.gather_rows(T(rows, dev), perm)
    assert np.array_equal(fwd.cpu().numpy(), rows[op])
    back = r.scatter_rows(fwd, perm)
    assert np.array_equal(back.cpu().numpy(), rows)
    mask = (keys & 1).astype(np.uint8)
    assert np.array_equal(r.scatter_rows(r.gather_rows(T(mask, dev), perm), perm).cpu().numpy(), mask)


def test_concurrent_insert_stress(dev):
    """Many waves claim slots in few buckets at once: no key stored twice, none lost (SURVEY §5 race row)."""
    dim, n = 4, 60000
    keys = synth.keys_np(19, 0, n)
    t = LookupTable(n + 64, dim, device=dev, max_batch=n)       # load ~1.0: heavy CAS contention and long probes
    rows = synth.rows_np(keys, dim, 1)
    t.insert(T(keys, dev), T(rows, dev))
    assert t.status() == 0 and t.size() == n
    gk, gv = t.export()
    gk = gk.cpu().numpy()
    assert len(np.unique(gk)) == n and np.array_equal(np.sort(gk), np.sort(keys))
    out, found = t.find(T(keys, dev))
    assert found.all() and np.array_equal(out.cpu().numpy(), rows)


def test_large_round_trip_properties(dev):
    """Size-independent properties at a bench-like size (8M keys, dim 64): insert -> find returns exactly the
    key-derived rows; absent stream all-miss; export is a permutation of the inserted set (checksum of keys)."""
    dim, n, chunk = 64, 8_000_000, 1_000_000
    t = LookupTable(int(n / 0.75), dim, device=dev, max_batch=chunk)
    for s in range(0, n, chunk):
        k = synth.keys_t(1, s, chunk, dev)
        t.insert(k, synth.rows_t(k, dim, 2))
    assert t.size() == n and t.status() == 0
    for s in (0, 3_000_000, 7_000_000):
        k = synth.keys_t(1, s, chunk, dev)
        out, found = t.find(k)
        assert bool(found.all()) and torch.equal(out, synth.rows_t(k, dim, 2))
    k = synth.keys_t(2, 0, chunk, dev)
    out, found = t.find(k)
    assert not bool(found.any()) and not bool(out.any())
    gk, gv = t.export()
    assert gk.numel() == n
    allk = torch.cat([synth.keys_t(1, s, chunk, dev) for s in range(0, n, chunk)])
    assert int(gk.sum()) == int(allk.sum()) and int((gk ^ (gk >> 7)).sum()) == int((allk ^ (allk >> 7)).sum())
    assert torch.equal(gv[:chunk], synth.rows_t(gk[:chunk], dim, 2))


def _free_hbm_gb(dev):
    free, _ = torch.cuda.mem_get_info(dev)
    return free / 1e9


def test_full_size_configs_properties(dev):
    """BASELINE configs[1]/[2] at their FULL size — 100M keys, dim 64, load 0.75, with an Adagrad plane (69 GB) — through size-independent
    properties: every inserted key is found with exactly its key-derived row (whole key stream, 1M-key batches), an absent stream misses
    everywhere, size() is exact, the export is a permutation of the inserted set (checksums of keys, rows re-derived from exported keys),
    and one sparse-Adagrad step with a key-derived gradient moves every row to the value the update formula gives (checked on samples,
    bit for bit: no duplicates -> no reduction-order freedom), on the located path and the probing path alike."""
    if _free_hbm_gb(dev) < 120:
        pytest.skip("needs ~100 GB of free HBM")
    dim, n, chunk, M64 = 64, 100_000_000, 1 << 20, 1 << 64
    t = LookupTable(int(n / 0.75), dim, device=dev, max_batch=chunk, optimizer=OPT_ADAGRAD, initial_accumulator=0.1)
    ksum = kmix = 0
    for s in range(0, n, chunk):
        k = synth.keys_t(1, s, min(chunk, n - s), dev)
        t.insert(k, synth.rows_t(k, dim, 2))
        ksum = (ksum + int(k.sum())) % M64; kmix = (kmix + int((k ^ (k >> 7)).sum())) % M64   # checksums mod 2^64, whatever the chunking
    assert t.size() == n and t.status() == 0
    for s in range(0, n, chunk):                      # the whole key stream
        k = synth.keys_t(1, s, min(chunk, n - s), dev)
        out, found = t.find(k)
        assert bool(found.all()) and torch.equal(out, synth.rows_t(k, dim, 2)), s
    k = synth.keys_t(2, 0, chunk, dev)
    out, found = t.find(k)
    assert not bool(found.any()) and not bool(out.any())
    # one Adagrad step over two 1M-key batches: located slots for the first, probing for the second
    lr, eps = 0.05, 1e-10
    for j, s in enumerate((5 * chunk, 77 * chunk)):
        k = synth.keys_t(1, s, chunk, dev)
        g = synth.rows_t(k, dim, 6) * 0.02
        w0 = synth.rows_t(k, dim, 2)
        if j == 0:
            _, _, slots = t.find_located(k)
            t.apply_adagrad(k, g, lr=lr, eps=eps, slots=slots)
        else:
            t.apply_adagrad(k, g, lr=lr, eps=eps)
        acc = torch.addcmul(torch.full_like(g, 0.1), g, g)            # fma(g, g, acc) — one rounding, like the kernel
        exp = torch.addcdiv(w0, g, acc.sqrt() + eps, value=-lr)        # w - lr * (g / (sqrt(acc) + eps)); compared at 1e-6: torch may fuse differently
        out, found = t.find(k)
        assert bool(found.all())
        torch.testing.assert_close(out, exp, rtol=1e-6, atol=1e-9)
        st, _ = t.find_plane(1, k[:4096])
        torch.testing.assert_close(st, acc[:4096], rtol=1e-6, atol=1e-12)
    # untouched keys kept their rows
    k = synth.keys_t(1, 40 * chunk, chunk, dev)
    out, _ = t.find(k)
    assert torch.equal(out, synth.rows_t(k, dim, 2))
    # export in slot ranges (bounded scratch): a permutation of the inserted keys
    n_exp = esum = emix = 0
    for ek, ev, e1, _ in t.iter_export(1 << 24, with_state=True):
        n_exp += ek.numel(); esum = (esum + int(ek.sum())) % M64; emix = (emix + int((ek ^ (ek >> 7)).sum())) % M64
    assert n_exp == n and esum == ksum and emix == kmix
    t.close()


# MEE_SOAK=N appends N more seeds (a soak run on the GPU box; the default suite keeps five); MEE_SOAK_FIRST=K: the N seeds start behind the first K (another soak, other sequences)
_SEQ = [(0, 16, "adagrad"), (1, 64, "adam"), (2, 128, "adagrad"), (3, 8, "adam"), (4, 64, "adagrad")] + \
       [(s, [16, 64, 128, 8][s % 4], ["adagrad", "adam"][(s // 4) % 2]) for s in range(5 + int(os.environ.get("MEE_SOAK_FIRST", "0")), 5 + int(os.environ.get("MEE_SOAK_FIRST", "0")) + int(os.environ.get("MEE_SOAK", "0")))]


@pytest.mark.parametrize("seed,dim,opt", _SEQ)
def test_random_op_sequences(dev, seed, dim, opt):
    """Differential test: a random sequence of every operator (skewed duplicate-heavy batches, reserved keys, absent
    keys, a table that runs close to full) on the HIP backend and on the oracle; all observables compared after each op."""
    rng = np.random.default_rng(1000 + seed)
    cap = ocap = 16 * 64
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    kw = dict(default_value=0.25, initial_accumulator=0.1, initializer=INIT_UNIFORM, init_scale=0.05, init_seed=seed)
    t = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=4096, **kw)
    o = oracle.OracleTable(cap, dim, optimizer=okind, **kw)
    universe = synth.keys_np(50 + seed, 0, 1400)      # more keys than slots: TABLE_FULL paths get exercised
    step = 0
    for it in range(60):
        n = int(rng.integers(1, 1500))
        idx = np.minimum(rng.zipf(1.2, size=n) - 1, universe.size - 1) if rng.random() < 0.5 else rng.integers(0, universe.size, n)
        keys = universe[idx].copy()
        if rng.random() < 0.2:
            keys[rng.integers(0, n)] = oracle.EMPTY_KEY if rng.random() < 0.5 else oracle.RECLAIMED_KEY
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        op = rng.choice(["insert", "assign", "remove", "find", "find_or_insert", "apply", "apply", "dedup", "reserve"])
        full_before = bool(o.status() & STATUS_TABLE_FULL)
        if op == "reserve":   # in-place rehash to a random capacity that still holds everything: no observable may change
            t.reserve(int(rng.integers(o.size() + 64, 2 * ocap)))
            cap = min(t.capacity, ocap)
        elif op == "insert":
            if o.size() + len(np.unique(keys)) > cap - 16:
                continue   # which keys get dropped on overflow is placement-order dependent; overflow has its own test
            t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
        elif op == "assign":
            assert np.array_equal(t.assign(T(keys, dev), T(rows, dev)).cpu().numpy(), o.assign(keys, rows))
        elif op == "remove":
            keys = keys[: max(1, n // 4)]
            assert np.array_equal(t.remove(T(keys, dev)).cpu().numpy(), o.remove(keys))
        elif op == "find":
            out, found = t.find(T(keys, dev)); eo, ef = o.find(keys)
            assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
        elif op == "find_or_insert":
            if o.size() + len(np.unique(keys)) > cap - 16:
                continue
            out, found = t.find_or_insert(T(keys, dev)); eo, ef = o.find_or_insert(keys)
            assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
        elif op == "apply":
            step += 1
            g = (rows * 0.01).astype(np.float32)
            if opt == "adagrad":
                t.apply_adagrad(T(keys, dev), T(g, dev), lr=0.02, eps=1e-10); o.apply_adagrad(keys, g, 0.02, 1e-10)
            else:
                t.apply_adam(T(keys, dev), T(g, dev), lr=0.002, step=step); o.apply_adam(keys, g, 0.002, 0.9, 0.999, 1e-8, step)
        else:
            uniq, gs, cnt, inv = t.dedup_sum(T(keys, dev), T(rows, dev), compact=True)
            ou, ogs, oinv, ocnt = oracle.dedup_sum(keys, rows, dim)
            a, b = np.argsort(uniq.cpu().numpy()), np.argsort(ou)
            assert np.array_equal(uniq.cpu().numpy()[a], ou[b]) and np.array_equal(cnt.cpu().numpy()[a], ocnt[b])
            np.testing.assert_allclose(gs.cpu().numpy()[a], ogs[b], rtol=RTOL, atol=ATOL)
        assert t.size() == o.size(), (it, op)
        assert bool(t.status() & STATUS_TABLE_FULL) == bool(o.status() & STATUS_TABLE_FULL) == full_before
    g_ = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
    o_ = o.export(with_state=True)
    a, b = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][a], o_[0][b])
    for x, y in zip(g_[1:], o_[1:]):
        if y is not None:
            np.testing.assert_allclose(x[a], y[b], rtol=RTOL, atol=ATOL)


def test_ops_are_graph_capturable(dev):
    """The header promises that the hot ops neither allocate nor synchronise: capture find + Adagrad apply + insert in a
    hipGraph (torch.cuda.graph on a side stream), replay it twice, and check the table against the oracle."""
    dim, n = 64, 5000
    keys = synth.keys_np(61, 0, n); rows = synth.rows_np(keys, dim, 2)
    extra = synth.keys_np(62, 0, 100); extra_rows = synth.rows_np(extra, dim, 3)
    g = (synth.rows_np(keys, dim, 6) * 0.02).astype(np.float32)
    t = LookupTable(16384, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=n); o = oracle.OracleTable(16384, dim, optimizer=oracle.OPT_ADAGRAD)
    dk, dr, dg, dek, der = T(keys, dev), T(rows, dev), T(g, dev), T(extra, dev), T(extra_rows, dev)
    t.insert(dk, dr); o.insert(keys, rows)
    out = torch.empty((n, dim), dtype=torch.float32, device=dev); found = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        t.find(dk, out=out, found=found)
        t.apply_adagrad(dk, dg, lr=0.01, eps=1e-10)
        t.insert(dek, der)
    for _ in range(2):
        eo, ef = o.find(keys)
        o.apply_adagrad(keys, g, 0.01, 1e-10); o.insert(extra, extra_rows)
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), eo) and np.array_equal(found.cpu().numpy(), ef)
    assert t.size() == o.size() == n + 100
    got, _ = t.find(dk); exp, _ = o.find(keys)
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("dim", [64, 128, 24])
def test_find_parity_under_every_tuning_knob(dev, dim):
    """Performance knobs must never change results: every rounds / cache-policy / grid-cap variant of find_kernel
    (and both apply_single variants) against the oracle, with misses, a reserved key and a ragged batch size."""
    n = 30011
    keys = synth.keys_np(81, 0, 2 * n); rows = synth.rows_np(keys[:n], dim, 2)
    t = LookupTable(int(n / 0.8), dim, device=dev, max_batch=2 * n, default_value=1.5, optimizer=OPT_ADAGRAD)
    o = oracle.OracleTable(int(n / 0.8), dim, default_value=1.5, optimizer=oracle.OPT_ADAGRAD)
    t.insert(T(keys[:n], dev), T(rows, dev)); o.insert(keys[:n], rows)
    q = keys[np.random.default_rng(0).permutation(2 * n)][: n + 7].copy()
    q[5] = oracle.RECLAIMED_KEY
    eo, ef = o.find(q)
    dq = T(q, dev)
    for rounds in (0, 1, 2, 4, 8):
        for nt in (-1, 0, 1, 2, 3, 4, 5, 6, 7):
            for cap in (0, 64):
                t.set_tuning("find_rounds", rounds); t.set_tuning("find_nt", nt); t.set_tuning("find_grid_cap", cap)
                out, found = t.find(dq)
                assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo), (rounds, nt, cap)
    with pytest.raises(MeepoError):
        t.set_tuning("no_such_knob", 1)
    g = (synth.rows_np(keys[:n], dim, 6) * 0.02).astype(np.float32)
    for retired in ("apply_rounds", "apply_path", "apply_overlap", "apply_spare_blocks", "dedup_path"):   # knobs of deleted code paths are unknown names now
        with pytest.raises(MeepoError):
            t.set_tuning(retired, 1)
    for kernel, bmax, split in ((0, 0, 0), (1, 128, 548), (-1, 352, 300)):      # the apply's knobs: never change results
        t.set_tuning("apply_kernel", kernel); t.set_tuning("apply_bucket_max", bmax); t.set_tuning("apply_xcd_split", split)
        t.apply_adagrad(T(keys[:n], dev), T(g, dev), lr=0.01); o.apply_adagrad(keys[:n], g, 0.01, 1e-10)
        out, _ = t.find(T(keys[:n], dev)); exp, _ = o.find(keys[:n])
        assert np.array_equal(out.cpu().numpy(), exp), (kernel, bmax, split)


@pytest.mark.parametrize("dim", [4, 512, 1024])
def test_extreme_row_widths(dev, dim):
    """Narrowest and widest rows through every row-moving kernel: insert, find, find_or_insert, Adam apply with
    duplicates (small and big groups), export with state, remove."""
    n = 3000
    kw = dict(initializer=INIT_UNIFORM, init_scale=0.1, init_seed=3)
    t = LookupTable(8192, dim, device=dev, optimizer=OPT_ADAM, max_batch=8192, **kw); o = oracle.OracleTable(8192, dim, optimizer=oracle.OPT_ADAM, **kw)
    keys = synth.keys_np(91, 0, n); rows = synth.rows_np(keys, dim, 2)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    rng = np.random.default_rng(dim)
    q = np.concatenate([keys[rng.integers(0, n, 4000)], synth.keys_np(92, 0, 500)])
    out, found = t.find_or_insert(T(q, dev)); eo, ef = o.find_or_insert(q)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    bk = np.concatenate([keys[np.minimum(rng.zipf(1.2, 5000) - 1, n - 1)], np.full(200, keys[7])])
    g = (rng.standard_normal((bk.size, dim)) * 0.01).astype(np.float32)
    for s in range(2):
        t.apply_adam(T(bk, dev), T(g, dev), lr=0.001, step=s + 1); o.apply_adam(bk, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
    assert np.array_equal(t.remove(T(keys[:100], dev)).cpu().numpy(), o.remove(keys[:100]))
    g_ = [x.cpu().numpy() for x in t.export(with_state=True)]; o_ = o.export(with_state=True)
    a, b = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][a], o_[0][b])
    for x, y in zip(g_[1:], o_[1:]):
        np.testing.assert_allclose(x[a], y[b], rtol=RTOL, atol=ATOL)


def test_apply_prepare_split(dev):
    """mee_apply_prepare on a side stream beside the forward find, then the apply with the same keys: same result as the
    one-call apply / the oracle; other group-table users are refused while it is pending; discard leaves things clean."""
    dim, n_keys, batch = 64, 20000, 30000
    keys = synth.keys_np(71, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    t = LookupTable(32768, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=batch, initial_accumulator=0.1)
    o = oracle.OracleTable(32768, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    rng = np.random.default_rng(9)
    side = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    for s in range(3):
        bk = keys[np.minimum(rng.zipf(1.3, size=batch) - 1, n_keys - 1)]
        g = (rng.standard_normal((batch, dim)) * 0.01).astype(np.float32)
        dk, dg = T(bk, dev), T(g, dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            t.apply_prepare(dk)                       # overlaps with the lookup below
        out, found = t.find(dk)
        eo, ef = o.find(bk)
        with pytest.raises(MeepoError):
            t.insert(dk[:10], torch.zeros(10, dim, device=dev))   # group table is busy
        with pytest.raises(MeepoError):
            t.remove(dk[:10])                                        # would lend out the scratch the prepared batch keeps its list heads in
        with pytest.raises(MeepoError):
            t.apply_adagrad(dk[:100], dg[:100], lr=0.01)             # not the prepared batch
        main.wait_stream(side)
        t.apply_adagrad(dk, dg, lr=0.01, eps=1e-10); o.apply_adagrad(bk, g, 0.01, 1e-10)
        assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    t.apply_prepare(T(keys[:5000], dev))
    t.apply_discard()
    # a discarded batch WITH hot keys (the grouping pass has already marked their occurrences as filed): the marks must be gone, or the
    # next apply would file positions of a batch that no longer exists
    hot = np.concatenate([np.repeat(keys[:3], 400), keys[100:5000]]); rng.shuffle(hot)
    t.apply_prepare(T(hot, dev))
    t.apply_discard()
    bk = keys[rng.integers(0, n_keys, 6000)]
    g = (rng.standard_normal((6000, dim)) * 0.01).astype(np.float32)
    t.apply_adagrad(T(bk, dev), T(g, dev), lr=0.01, eps=1e-10); o.apply_adagrad(bk, g, 0.01, 1e-10)
    t.insert(T(keys[:10], dev), T(rows[:10], dev)); o.insert(keys[:10], rows[:10])   # accepted again, scratch is clean
    u, _, c, _ = t.dedup_sum(T(np.concatenate([keys[:100], keys[:50]]), dev), compact=True)
    assert u.numel() == 100 and int(c.sum()) == 150
    g_ = [x.cpu().numpy() for x in t.export(with_state=True)[:3]]; o_ = o.export(with_state=True)[:3]
    a, b = np.argsort(g_[0]), np.argsort(o_[0])
    assert np.array_equal(g_[0][a], o_[0][b])
    np.testing.assert_allclose(g_[1][a], o_[1][b], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(g_[2][a], o_[2][b], rtol=RTOL, atol=ATOL)


def test_export_import_roundtrip_with_state(dev):
    """export(with_state) -> import_ into a fresh table of another capacity (chunked by max_batch): identical table."""
    dim, n = 32, 25000
    keys = synth.keys_np(95, 0, n); rows = synth.rows_np(keys, dim, 2)
    a = LookupTable(40000, dim, device=dev, optimizer=OPT_ADAM, max_batch=n)
    a.insert(T(keys, dev), T(rows, dev))
    g = (synth.rows_np(keys, dim, 6) * 0.02).astype(np.float32)
    for s in range(2):
        a.apply_adam(T(keys, dev), T(g, dev), lr=0.001, step=s + 1)
    ek, ev, e1, e2 = a.export(with_state=True)
    b = LookupTable(65536, dim, device=dev, optimizer=OPT_ADAM, max_batch=4096)   # forces 7 chunks
    b.import_(ek, ev, e1, e2)
    assert b.size() == n
    for plane in (0, 1, 2):
        pa, fa = a.find_plane(plane, T(keys, dev)); pb, fb = b.find_plane(plane, T(keys, dev))
        assert bool(fa.all()) and bool(fb.all()) and torch.equal(pa, pb)
    a.apply_adam(T(keys, dev), T(g, dev), lr=0.001, step=3); b_keys = T(keys[:4096], dev)
    b.apply_adam(b_keys, T(g[:4096], dev), lr=0.001, step=3)       # training continues identically from the checkpoint
    assert torch.equal(a.find(b_keys)[0], b.find(b_keys)[0])


def test_resized_rehash(dev):
    """Growing = an explicit rehash into a new table: a table filled to load 0.95 (long probe chains) is rehashed to load
    0.5; every key, row and Adagrad accumulator survives and lookups agree with the oracle."""
    dim, cap = 16, 16 * 512
    n = int(cap * 0.95)
    keys = synth.keys_np(97, 0, n); rows = synth.rows_np(keys, dim, 2)
    t = LookupTable(cap, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=n, initial_accumulator=0.1)
    o = oracle.OracleTable(cap, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    g = (synth.rows_np(keys, dim, 6) * 0.02).astype(np.float32)
    t.apply_adagrad(T(keys, dev), T(g, dev), lr=0.01); o.apply_adagrad(keys, g, 0.01, 1e-10)
    big = t.resized(2 * cap)
    assert big.capacity >= 2 * cap and big.size() == n and big.status() == 0
    q = np.concatenate([keys, synth.keys_np(98, 0, 100)])
    out, found = big.find(T(q, dev)); eo, ef = o.find(q)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
    acc, _ = big.find_plane(1, T(keys, dev)); acc0, _ = t.find_plane(1, T(keys, dev))
    assert torch.equal(acc, acc0)


@pytest.mark.parametrize("dim", [64, 128, 24])
def test_grouped_find_equals_per_table_find(dev, dim):
    """mee_find_grouped: one launch over the concatenated batches of many tables == find per table (and == the oracle),
    with ragged and empty segments, reserved keys, absent keys, different capacities / default rows, and a member that is
    rehashed between calls."""
    from meepoembedding_amd import TableGroup
    rng = np.random.default_rng(dim)
    n_tables = 7
    tables, oracles, universes = [], [], []
    for j in range(n_tables):
        cap = int(rng.integers(200, 6000))
        t = LookupTable(cap, dim, device=dev, max_batch=4096, default_value=float(j))
        o = oracle.OracleTable(cap, dim, default_value=float(j))
        u = synth.keys_np(300 + j, 0, int(cap * 0.7))
        rows = rng.standard_normal((u.size, dim)).astype(np.float32)
        for s in range(0, u.size, 4096):
            t.insert(T(u[s:s + 4096], dev), T(rows[s:s + 4096], dev))
        o.insert(u, rows)
        tables.append(t); oracles.append(o); universes.append(u)
    grp = TableGroup(tables)
    for trial in range(4):
        lens = [int(rng.integers(0, 3000)) if rng.random() > 0.25 else 0 for _ in range(n_tables)]
        if trial == 3:
            lens = [0] * n_tables; lens[4] = 1     # a single key in the whole batch
        segs = []
        for j, m in enumerate(lens):
            k = universes[j][rng.integers(0, universes[j].size, m)].copy()
            if m > 4:
                k[1] = oracle.EMPTY_KEY; k[2] = synth.keys_np(999, j, 1)[0]; k[3] = universes[(j + 1) % n_tables][0]  # another table's key
            segs.append(k)
        keys = np.concatenate(segs) if sum(lens) else np.zeros(0, np.int64)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=dev)
        out, found = grp.find(T(keys, dev), offs)
        out, found = out.cpu().numpy(), found.cpu().numpy()
        p = 0
        for j, k in enumerate(segs):
            eo, ef = oracles[j].find(k)
            to, tf = tables[j].find(T(k, dev)) if k.size else (torch.zeros(0, dim), torch.zeros(0, dtype=torch.uint8))
            assert np.array_equal(found[p:p + k.size], ef) and np.array_equal(out[p:p + k.size], eo)
            assert np.array_equal(to.cpu().numpy(), eo) and np.array_equal(tf.cpu().numpy(), ef)
            p += k.size
        if trial == 1:
            tables[2].reserve(tables[2].capacity * 3)   # planes move: the group must pick up the new descriptors
    # offsets that do not start at 0 / end before n: positions outside are left untouched
    keys = np.concatenate([universes[0][:10], universes[1][:10]])
    sentinel = torch.full((20, dim), -7.0, device=dev)
    fsent = torch.full((20,), 9, dtype=torch.uint8, device=dev)
    offs = torch.tensor([3, 10] + [10] * (n_tables - 2) + [15], dtype=torch.int64, device=dev)   # table 0: [3,10), last table: [10,15)
    grp.find(T(keys, dev), offs, out=sentinel, found=fsent)
    assert bool((sentinel[:3] == -7).all()) and bool((sentinel[15:] == -7).all()) and bool((fsent[:3] == 9).all())
    assert np.array_equal(sentinel[3:10].cpu().numpy(), oracles[0].find(keys[3:10])[0])
    assert np.array_equal(sentinel[10:15].cpu().numpy(), oracles[n_tables - 1].find(keys[10:15])[0])
    with pytest.raises(MeepoError):
        TableGroup([tables[0], LookupTable(100, dim + 4, device=dev)])
    grp.close()


@pytest.mark.parametrize("opt,dim", [("adagrad", 64), ("adam", 128), ("adagrad", 24)])
def test_grouped_apply_equals_per_table_apply(dev, opt, dim):
    """mee_group_apply_*: one optimizer step over the jagged batch of a group == apply per table == the oracle, with
    duplicate-heavy segments (hot keys beyond the chunk and big-group thresholds), absent and reserved keys, the SAME key
    value stored in several tables, and empty segments."""
    from meepoembedding_amd import TableGroup
    rng = np.random.default_rng(7 + dim)
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    n_tables = 5
    shared = synth.keys_np(555, 0, 300)          # keys present in every table (different rows per table)
    grouped, solo, oracles, universes = [], [], [], []
    for j in range(n_tables):
        cap = int(rng.integers(1500, 5000))
        u = np.concatenate([shared, synth.keys_np(600 + j, 0, int(cap * 0.5))])
        rows = rng.standard_normal((u.size, dim)).astype(np.float32)
        kw = dict(initial_accumulator=0.1)
        a = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=8192, **kw)
        b = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=8192, **kw)
        o = oracle.OracleTable(cap, dim, optimizer=okind, **kw)
        a.insert(T(u, dev), T(rows, dev)); b.insert(T(u, dev), T(rows, dev)); o.insert(u, rows)
        grouped.append(a); solo.append(b); oracles.append(o); universes.append(u)
    grp = TableGroup(grouped, max_apply_batch=1 << 15)
    for step in range(1, 4):
        segs = []
        for j in range(n_tables):
            m = 0 if (j + step) % 4 == 0 else int(rng.integers(1, 4000))
            u = universes[j]
            idx = np.minimum(rng.zipf(1.15, size=m) - 1, u.size - 1) if m else np.zeros(0, np.int64)
            k = u[idx].copy()
            if m > 200:
                k[:100] = u[5]                                # one key 100+ times: beyond kChunk -> partial rows + tree
                k[100:103] = [oracle.EMPTY_KEY, oracle.RECLAIMED_KEY, synth.keys_np(998, j, 1)[0]]   # padding, reserved, absent
            segs.append(k)
        keys = np.concatenate(segs)
        grads = (rng.standard_normal((keys.size, dim)) * 0.05).astype(np.float32)
        offs = torch.tensor(np.concatenate([[0], np.cumsum([s.size for s in segs])]), dtype=torch.int64, device=dev)
        if opt == "adagrad":
            grp.apply_adagrad(T(keys, dev), offs, T(grads, dev), lr=0.05, eps=1e-10)
        else:
            grp.apply_adam(T(keys, dev), offs, T(grads, dev), lr=0.01, step=step)
        p = 0
        for j, k in enumerate(segs):
            gseg = grads[p:p + k.size]; p += k.size
            if not k.size:
                continue
            if opt == "adagrad":
                solo[j].apply_adagrad(T(k, dev), T(gseg, dev), lr=0.05, eps=1e-10); oracles[j].apply_adagrad(k, gseg, 0.05, 1e-10)
            else:
                solo[j].apply_adam(T(k, dev), T(gseg, dev), lr=0.01, step=step); oracles[j].apply_adam(k, gseg, 0.01, 0.9, 0.999, 1e-8, step)
    for j in range(n_tables):
        ga = [x.cpu().numpy() for x in grouped[j].export(with_state=True) if x is not None]
        sa = [x.cpu().numpy() for x in solo[j].export(with_state=True) if x is not None]
        oa = [x for x in oracles[j].export(with_state=True) if x is not None]
        ia, ib, ic = np.argsort(ga[0]), np.argsort(sa[0]), np.argsort(oa[0])
        assert np.array_equal(ga[0][ia], sa[0][ib]) and np.array_equal(ga[0][ia], oa[0][ic])
        for x, y, z in zip(ga[1:], sa[1:], oa[1:]):
            np.testing.assert_allclose(x[ia], y[ib], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(x[ia], z[ic], rtol=RTOL, atol=ATOL)
        assert grouped[j].status() == solo[j].status() == oracles[j].status()
    # errors: a group without apply scratch, a batch beyond max_apply_batch, mixed optimizers
    with pytest.raises(MeepoError):
        TableGroup(grouped).apply_adagrad(T(keys, dev), offs, T(grads, dev), lr=0.1)
    with pytest.raises(MeepoError):
        TableGroup(grouped, max_apply_batch=16).apply_adagrad(T(keys, dev), offs, T(grads, dev), lr=0.1)
    with pytest.raises(MeepoError):
        TableGroup([grouped[0], LookupTable(100, dim, device=dev)], max_apply_batch=64)
    grp.close()


@pytest.mark.parametrize("opt,dim", [("adagrad", 64), ("adam", 24), ("none", 128)])
def test_grouped_find_or_insert_equals_per_table(dev, opt, dim):
    """mee_group_find_or_insert == find_or_insert per table == the oracle: rows, present-before masks, the created keys'
    hashed initial rows and optimizer state, duplicates of a new key, reserved keys, and TABLE_FULL on the right member."""
    from meepoembedding_amd import OPT_NONE, TableGroup
    rng = np.random.default_rng(31 + dim)
    kind, okind = {"adagrad": (OPT_ADAGRAD, oracle.OPT_ADAGRAD), "adam": (OPT_ADAM, oracle.OPT_ADAM), "none": (OPT_NONE, oracle.OPT_NONE)}[opt]
    n_tables = 4
    grouped, oracles, universes = [], [], []
    for j in range(n_tables):
        cap = 4096 if j else 16 * 8            # member 0 is tiny: it overflows
        kw = dict(default_value=0.5 * j, initial_accumulator=0.1 * (j + 1), initializer=INIT_UNIFORM, init_scale=0.05, init_seed=70 + j)
        a = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=8192, **kw)
        o = oracle.OracleTable(cap, dim, optimizer=okind, **kw)
        u = synth.keys_np(800 + j, 0, 3000)
        rows = rng.standard_normal((60, dim)).astype(np.float32)
        a.insert(T(u[:60], dev), T(rows, dev)); o.insert(u[:60], rows)
        grouped.append(a); oracles.append(o); universes.append(u)
    grp = TableGroup(grouped, max_apply_batch=1 << 14)
    for trial in range(3):
        segs = []
        for j in range(n_tables):
            m = int(rng.integers(0, 1500)) if j else 40
            hi = 1000 * (trial + 1) if j else 100      # member 0: at most 100 distinct keys ever asked -> within its 128 slots
            k = universes[j][rng.integers(0, hi, m)].copy()
            if m > 10:
                k[3] = oracle.EMPTY_KEY; k[4] = oracle.RECLAIMED_KEY; k[5] = k[6]    # padding, reserved, a duplicate
            segs.append(k)
        keys = np.concatenate(segs)
        offs = torch.tensor(np.concatenate([[0], np.cumsum([s.size for s in segs])]), dtype=torch.int64, device=dev)
        out, found = grp.find_or_insert(T(keys, dev), offs)
        out, found = out.cpu().numpy(), found.cpu().numpy()
        p = 0
        for j, k in enumerate(segs):
            eo, ef = oracles[j].find_or_insert(k)
            assert np.array_equal(found[p:p + k.size], ef), (trial, j)
            assert np.array_equal(out[p:p + k.size], eo), (trial, j)
            p += k.size
    for j in range(n_tables):
        ga = [x.cpu().numpy() for x in grouped[j].export(with_state=True) if x is not None]
        oa = [x for x in oracles[j].export(with_state=True) if x is not None]
        ia, ic = np.argsort(ga[0]), np.argsort(oa[0])
        assert np.array_equal(ga[0][ia], oa[0][ic])
        for x, z in zip(ga[1:], oa[1:]):
            assert np.array_equal(x[ia], z[ic])
        assert grouped[j].status() == oracles[j].status() == STATUS_RESERVED_KEY   # every segment carried a RECLAIMED key
        grouped[j].clear_status()
    # overflow lands on the member that is full, and only there
    flood = universes[0][200:600]
    offs = torch.tensor([0, flood.size] + [flood.size] * (n_tables - 1), dtype=torch.int64, device=dev)
    _, f2 = grp.find_or_insert(T(flood, dev), offs)
    assert not bool(f2.any())
    assert grouped[0].status() & STATUS_TABLE_FULL and all(grouped[j].status() == 0 for j in range(1, n_tables))
    assert grouped[0].size() <= grouped[0].capacity
    grp.close()


@pytest.mark.parametrize("dim,mode", [(64, "sum"), (128, "mean"), (24, "sum"), (1024, "mean")])
def test_find_pooled_bit_exact(dev, dim, mode):
    """mee_find_pooled == segment-sum (position order, fp32) of find's rows: bit-exact, with empty bags, bags of one, long
    bags, absent and reserved keys, odd bag lengths (the kernel keeps two keys in flight)."""
    rng = np.random.default_rng(dim)
    n_keys = 3000
    t = LookupTable(6000, dim, device=dev, max_batch=4096, default_value=0.125)
    o = oracle.OracleTable(6000, dim, default_value=0.125)
    u = synth.keys_np(41, 0, n_keys)
    rows = rng.standard_normal((n_keys, dim)).astype(np.float32)
    t.insert(T(u, dev), T(rows, dev)); o.insert(u, rows)
    # two launch shapes: mostly short bags (a tile per bag, long ones shared by the wave) and a long average (a wave per bag)
    for lens in (np.concatenate([[0, 1, 2, 3, 0, 57, 400, 1], rng.integers(0, 12, 500), [0]]),
                 np.concatenate([[0, 1, 33], rng.integers(5, 60, 120), [0, 16, 15, 17]])):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        keys = u[rng.integers(0, n_keys, off[-1])].copy()
        keys[5] = oracle.EMPTY_KEY; keys[9] = synth.keys_np(777, 0, 1)[0]; keys[30] = oracle.RECLAIMED_KEY
        out, found = t.find_pooled(T(keys, dev), T(off, dev), mode)
        er, ef = o.find(keys)
        assert np.array_equal(found.cpu().numpy(), ef)
        assert np.array_equal(out.cpu().numpy(), oracle.pool_rows(er, off, mode))
    # the unpooled find of the same keys, pooled by torch in fp64, agrees to rounding (sanity of the oracle helper itself)
    ref = torch.zeros(lens.size, dim, dtype=torch.float64).index_add_(0, torch.repeat_interleave(torch.arange(lens.size), torch.from_numpy(lens)), torch.from_numpy(er).double())
    if mode == "mean":
        ref = ref / torch.from_numpy(np.maximum(lens, 1)).double()[:, None]
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    e_out, _ = t.find_pooled(T(keys[:0], dev), torch.zeros(1, dtype=torch.int64, device=dev))   # zero bags
    assert e_out.shape == (0, dim)
    e_out, _ = t.find_pooled(T(keys[:0], dev), torch.zeros(4, dtype=torch.int64, device=dev))   # three bags, all empty
    assert e_out.shape == (3, dim) and not bool(e_out.any())


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_indexed_apply_is_apply_of_gathered_grads(dev, opt):
    """apply_*_indexed(keys, bag_grads, bag_of_position) == apply_*(keys, bag_grads[bag_of_position]) == the oracle."""
    rng = np.random.default_rng(3)
    dim, n_keys = 64, 2000
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    a = LookupTable(4096, dim, device=dev, optimizer=kind, max_batch=1 << 14, initial_accumulator=0.1)
    b = LookupTable(4096, dim, device=dev, optimizer=kind, max_batch=1 << 14, initial_accumulator=0.1)
    o = oracle.OracleTable(4096, dim, optimizer=okind, initial_accumulator=0.1)
    u = synth.keys_np(43, 0, n_keys)
    rows = rng.standard_normal((n_keys, dim)).astype(np.float32)
    for x in (a, b):
        x.insert(T(u, dev), T(rows, dev))
    o.insert(u, rows)
    for step in (1, 2):
        lens = rng.integers(0, 9, 1500)
        n = int(lens.sum())
        idx = np.minimum(rng.zipf(1.2, n) - 1, n_keys - 1)
        keys = u[idx].copy(); keys[:80] = u[7]           # a hot key: chunked and tree-summed paths read through the index too
        bag_grads = (rng.standard_normal((lens.size, dim)) * 0.05).astype(np.float32)
        bag_of = np.repeat(np.arange(lens.size), lens).astype(np.int64)
        if opt == "adagrad":
            a.apply_adagrad(T(keys, dev), T(bag_grads, dev), lr=0.05, grad_index=T(bag_of, dev))
            b.apply_adagrad(T(keys, dev), T(bag_grads[bag_of], dev), lr=0.05)
            o.apply_adagrad(keys, bag_grads[bag_of], 0.05, 1e-10)
        else:
            a.apply_adam(T(keys, dev), T(bag_grads, dev), lr=0.01, step=step, grad_index=T(bag_of, dev))
            b.apply_adam(T(keys, dev), T(bag_grads[bag_of], dev), lr=0.01, step=step)
            o.apply_adam(keys, bag_grads[bag_of], 0.01, 0.9, 0.999, 1e-8, step)
    ea, eb = a.export(with_state=True), b.export(with_state=True)
    eo = o.export(with_state=True)
    ia, ib, io = torch.argsort(ea[0]).cpu(), torch.argsort(eb[0]).cpu(), np.argsort(eo[0])
    for x, y, z in zip(ea[1:], eb[1:], eo[1:]):
        if x is not None:
            np.testing.assert_allclose(x.cpu()[ia].numpy(), y.cpu()[ib].numpy(), rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(x.cpu()[ia].numpy(), z[io], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("dim,mode,opt", [(64, "sum", "adagrad"), (128, "mean", "adam"), (24, "sum", "adagrad")])
def test_group_pooled_equals_per_table_pooled(dev, dim, mode, opt):
    """The embedding-bag collection: mee_group_find_pooled == find_pooled per member (bit-exact), and its backward
    (mee_group_apply_*_pooled) == apply_*_indexed per member == the oracle."""
    from meepoembedding_amd import TableGroup
    rng = np.random.default_rng(dim + 1)
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    n_tables, bpt = 4, 37
    a, b, o, univ = [], [], [], []
    for j in range(n_tables):
        cap = int(rng.integers(1000, 4000))
        u = synth.keys_np(900 + j, 0, int(cap * 0.6))
        rows = rng.standard_normal((u.size, dim)).astype(np.float32)
        kw = dict(default_value=0.25 * j, initial_accumulator=0.1)
        x = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=1 << 14, **kw)
        y = LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=1 << 14, **kw)
        z = oracle.OracleTable(cap, dim, optimizer=okind, **kw)
        x.insert(T(u, dev), T(rows, dev)); y.insert(T(u, dev), T(rows, dev)); z.insert(u, rows)
        a.append(x); b.append(y); o.append(z); univ.append(u)
    grp = TableGroup(a, max_apply_batch=1 << 14)
    for step, long_bags in ((1, False), (2, True)):          # both launch shapes of the pooled kernel
        lens = rng.integers(8, 40, n_tables * bpt) if long_bags else rng.integers(0, 7, n_tables * bpt)
        lens[3] = 0; lens[bpt] = 25
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        segs = []
        for j in range(n_tables):
            m = int(off[(j + 1) * bpt] - off[j * bpt])
            k = univ[j][np.minimum(rng.zipf(1.3, m) - 1, univ[j].size - 1)].copy()
            if m > 3:
                k[1] = synth.keys_np(997, j, 1)[0]          # absent
            segs.append(k)
        keys = np.concatenate(segs)
        located = torch.empty(keys.size, dtype=torch.int64, device=dev) if long_bags else None   # second step: the forward hands its located rows over
        out, found = grp.find_pooled(T(keys, dev), T(off, dev), mode, located=located)
        out, found = out.cpu().numpy(), found.cpu().numpy()
        for j in range(n_tables):
            lo, hi = off[j * bpt], off[(j + 1) * bpt]
            er, ef = o[j].find(keys[lo:hi])
            assert np.array_equal(found[lo:hi], ef)
            assert np.array_equal(out[j * bpt:(j + 1) * bpt], oracle.pool_rows(er, off[j * bpt:(j + 1) * bpt + 1] - lo, mode))
        bag_grads = (rng.standard_normal((n_tables * bpt, dim)) * 0.05).astype(np.float32)
        bag_of = np.repeat(np.arange(lens.size), lens).astype(np.int64)
        kwargs = dict(lr=0.05) if opt == "adagrad" else dict(lr=0.01, step=step)
        grp.apply_pooled(T(keys, dev), T(off, dev), T(bag_grads, dev), T(bag_of, dev), opt, located=located, **kwargs)
        for j in range(n_tables):
            lo, hi = off[j * bpt], off[(j + 1) * bpt]
            if hi == lo:
                continue
            kj, gj = keys[lo:hi], bag_grads[bag_of[lo:hi]]
            if opt == "adagrad":
                b[j].apply_adagrad(T(kj, dev), T(bag_grads, dev), lr=0.05, grad_index=T(bag_of[lo:hi], dev)); o[j].apply_adagrad(kj, gj, 0.05, 1e-10)
            else:
                b[j].apply_adam(T(kj, dev), T(bag_grads, dev), lr=0.01, step=step, grad_index=T(bag_of[lo:hi], dev)); o[j].apply_adam(kj, gj, 0.01, 0.9, 0.999, 1e-8, step)
    for j in range(n_tables):
        ga = [x.cpu().numpy() for x in a[j].export(with_state=True) if x is not None]
        gb = [x.cpu().numpy() for x in b[j].export(with_state=True) if x is not None]
        go = [x for x in o[j].export(with_state=True) if x is not None]
        ia, ib, io = np.argsort(ga[0]), np.argsort(gb[0]), np.argsort(go[0])
        for x, y, z in zip(ga[1:], gb[1:], go[1:]):
            np.testing.assert_allclose(x[ia], y[ib], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(x[ia], z[io], rtol=RTOL, atol=ATOL)
    grp.close()


def test_find_pooled_tolerates_bad_offsets(dev):
    """Offsets are caller data: bags that run past the key array are cut at its end, a decreasing pair is an empty bag —
    nothing is read out of bounds."""
    dim = 64
    t = LookupTable(1000, dim, device=dev, max_batch=1024)
    keys = synth.keys_np(50, 0, 100)
    rows = np.ones((100, dim), np.float32)
    t.insert(T(keys, dev), T(rows, dev))
    off = torch.tensor([0, 10, 5, 90, 1 << 40, 1 << 41], dtype=torch.int64, device=dev)   # [0,10) [10,5)=empty [5,90) [90,2^40)->[90,100) [2^40,2^41)=empty
    out, _ = t.find_pooled(T(keys, dev), off, "sum")
    assert out[:, 0].tolist() == [10.0, 0.0, 85.0, 10.0, 0.0]


def test_indexed_apply_clamps_bad_indices(dev):
    """grad_index is caller data: an index past the grad array is clamped to its last row instead of reading out of bounds."""
    dim = 64
    a = LookupTable(1000, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=1024)
    b = LookupTable(1000, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=1024)
    keys = synth.keys_np(51, 0, 200)
    for t in (a, b):
        t.insert(T(keys, dev), torch.zeros(200, dim, device=dev))
    grads = torch.randn(10, dim, device=dev)
    idx = torch.randint(0, 10, (200,), device=dev)
    bad = idx.clone(); bad[::7] = 1 << 30
    a.apply_adagrad(T(keys, dev), grads, lr=0.1, grad_index=bad)
    b.apply_adagrad(T(keys, dev), grads, lr=0.1, grad_index=torch.where(bad >= 10, torch.full_like(bad, 9), bad))
    assert torch.equal(a.find(T(keys, dev))[0], b.find(T(keys, dev))[0])


def test_group_and_pooled_ops_are_graph_capturable(dev):
    """The grouped / pooled entry points neither allocate nor synchronise either: capture a collection step (pooled
    lookup with located hand-over + grouped Adagrad + a find_or_insert on one member) in a hipGraph, replay it three
    times with host work and D2H copies in between, and compare with an eager twin."""
    from meepoembedding_amd import TableGroup
    rng = np.random.default_rng(77)
    dim, n_tables, bpt = 64, 3, 50
    mk = lambda: [LookupTable(4096, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=8192, initial_accumulator=0.1) for _ in range(n_tables)]
    a, b = mk(), mk()
    univ = [synth.keys_np(130 + j, 0, 1500) for j in range(n_tables)]
    for j in range(n_tables):
        rows = rng.standard_normal((1500, dim)).astype(np.float32)
        a[j].insert(T(univ[j], dev), T(rows, dev)); b[j].insert(T(univ[j], dev), T(rows, dev))
    ga, gb = TableGroup(a, max_apply_batch=8192), TableGroup(b, max_apply_batch=8192)
    lens = rng.integers(0, 9, n_tables * bpt)
    off = T(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), dev)
    keys = T(np.concatenate([univ[j][rng.integers(0, 1500, int(lens[j * bpt:(j + 1) * bpt].sum()))] for j in range(n_tables)]), dev)
    bag_of = T(np.repeat(np.arange(lens.size), lens).astype(np.int64), dev)
    grads = T((rng.standard_normal((n_tables * bpt, dim)) * 0.05).astype(np.float32), dev)
    fresh = T(synth.keys_np(140, 0, 64), dev)
    n = keys.numel()
    out = torch.empty((n_tables * bpt, dim), device=dev); found = torch.empty(n, dtype=torch.uint8, device=dev)
    located = torch.empty(n, dtype=torch.int64, device=dev)
    fo = torch.empty((64, dim), device=dev); ff = torch.empty(64, dtype=torch.uint8, device=dev)

    def step(grp, tables, o, f, loc, fo_, ff_):
        grp.find_pooled(keys, off, "sum", out=o, found=f, located=loc)
        grp.apply_pooled(keys, off, grads, bag_of, "adagrad", lr=0.05, located=loc)
        tables[1].find_or_insert(fresh, out=fo_, found=ff_)

    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(ga, a, out, found, located, fo, ff)
    out_b = torch.empty_like(out); found_b = torch.empty_like(found); loc_b = torch.empty_like(located)
    fo_b = torch.empty_like(fo); ff_b = torch.empty_like(ff)
    for it in range(3):
        step(gb, b, out_b, found_b, loc_b, fo_b, ff_b)
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), out_b.cpu().numpy()) and np.array_equal(ff.cpu().numpy(), ff_b.cpu().numpy())
        assert bool(ff.all()) == (it > 0)          # the fresh keys exist from the second step on
    for x, y in zip(a, b):
        ex, ey = x.export(with_state=True), y.export(with_state=True)
        ix, iy = torch.argsort(ex[0]), torch.argsort(ey[0])
        assert torch.equal(ex[0][ix], ey[0][iy])
        np.testing.assert_allclose(ex[1][ix].cpu().numpy(), ey[1][iy].cpu().numpy(), rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(ex[2][ix].cpu().numpy(), ey[2][iy].cpu().numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("fused", [False, True], ids=["separate", "forward_carries_partition"])
def test_train_step_is_graph_capturable(dev, fused):
    """One table's training step — find_or_insert_located (fused: its launch also carries the apply's partition) + the located apply, on a
    batch with duplicate groups of every kind (runs of 3 / 20 / 100 / 900: chunks, partial rows, a split bucket) — captured in a hipGraph and
    replayed: the apply's scratch (bucket totals and their two copies, tickets, pending counters) must be left clean by every replay, whatever
    state the capture froze into the kernel arguments.  Compared with an eager twin."""
    rng = np.random.default_rng(5)
    dim, n_keys = 64, 3000
    keys = synth.keys_np(150, 0, n_keys)
    mk = lambda: LookupTable(8192, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=16384, initial_accumulator=0.1,
                             initializer=INIT_UNIFORM, init_scale=0.05, init_seed=3)
    a, b = mk(), mk()
    rows = rng.standard_normal((2000, dim)).astype(np.float32)
    for t in (a, b):
        t.insert(T(keys[:2000], dev), T(rows, dev))
    reps = np.concatenate([np.ones(1500, int), np.full(200, 3), np.full(40, 20), np.full(6, 100), [900]])
    batch = np.repeat(keys[:reps.size], reps); rng.shuffle(batch)
    batch = np.concatenate([batch, keys[2000:2300]])          # unseen ids: created by the forward
    kb = T(batch, dev)
    n = batch.size
    g = T((rng.standard_normal((n, dim)) * 0.02).astype(np.float32), dev)
    bufs = lambda: (torch.empty((n, dim), device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.int64, device=dev))
    oa, fa, sa = bufs(); ob, fb, sb = bufs()

    def step(t, o, f, s_):
        t.find_or_insert_located(kb, out=o, found=f, slots=s_, prepare_apply=fused)
        t.apply_adagrad(kb, g, lr=0.05, slots=s_)

    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(a, oa, fa, sa)
    for it in range(3):
        step(b, ob, fb, sb)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and torch.equal(fa, fb)
    a.apply_adagrad(kb[:500], g[:500], lr=0.05); b.apply_adagrad(kb[:500], g[:500], lr=0.05)   # eager calls after the replays
    ea, eb = a.export(with_state=True), b.export(with_state=True)
    ia, ib = torch.argsort(ea[0]), torch.argsort(eb[0])
    assert torch.equal(ea[0][ia], eb[0][ib]) and a.status() == 0
    np.testing.assert_allclose(ea[1][ia].cpu().numpy(), eb[1][ib].cpu().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ea[2][ia].cpu().numpy(), eb[2][ib].cpu().numpy(), rtol=RTOL, atol=ATOL)


_GSEQ = list(range(4)) + list(range(4, 4 + int(os.environ.get("MEE_SOAK_GROUPS", "0"))))


@pytest.mark.parametrize("seed", _GSEQ)
def test_random_group_sequences(dev, seed):
    """Differential test of the grouped / pooled entry points: a random sequence of grouped find, find_or_insert, apply,
    pooled lookup and pooled apply (with and without the located hand-over) over a random collection of tables, interleaved
    with single-table ops on the members (insert, remove, reserve); every observable against one oracle table per member."""
    from meepoembedding_amd import TableGroup
    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.choice([8, 24, 64, 128]))
    opt = "adagrad" if seed % 2 == 0 else "adam"
    kind, okind = (OPT_ADAGRAD, oracle.OPT_ADAGRAD) if opt == "adagrad" else (OPT_ADAM, oracle.OPT_ADAM)
    n_tables = int(rng.integers(1, 6))
    tabs, orcs, univ = [], [], []
    for j in range(n_tables):
        cap = int(rng.integers(600, 3000))
        kw = dict(default_value=0.125 * j, initial_accumulator=0.1, initializer=INIT_UNIFORM, init_scale=0.05, init_seed=seed * 10 + j)
        tabs.append(LookupTable(cap, dim, device=dev, optimizer=kind, max_batch=1 << 14, **kw))
        orcs.append(oracle.OracleTable(1 << 14, dim, optimizer=okind, **kw))
        univ.append(synth.keys_np(7000 + seed * 10 + j, 0, 400))
    grp = TableGroup(tabs, max_apply_batch=1 << 14)
    step = 0

    def batch(pooled):
        bpt = int(rng.integers(1, 12)) if pooled else 1
        lens = rng.integers(0, 30 if rng.random() < 0.3 else 6, n_tables * bpt) if pooled else rng.integers(0, 300, n_tables)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        segs = []
        for j in range(n_tables):
            m = int(off[(j + 1) * bpt] - off[j * bpt])
            k = univ[j][np.minimum(rng.zipf(1.2, m) - 1, 399)].copy() if rng.random() < 0.5 else univ[j][rng.integers(0, 400, m)].copy()
            if m > 6:
                k[2] = oracle.EMPTY_KEY; k[4] = oracle.RECLAIMED_KEY
            segs.append(k)
        return bpt, lens, off, segs, (np.concatenate(segs) if off[-1] else np.zeros(0, np.int64))

    for it in range(30):
        op = rng.choice(["find", "foi", "apply", "pooled", "papply", "insert", "remove", "reserve"])
        if op in ("find", "foi", "apply"):
            _, lens, off, segs, keys = batch(False)
            if keys.size == 0:
                continue
            d_off = T(off, dev)
            if op == "find":
                out, found = grp.find(T(keys, dev), d_off)
                exp = [orcs[j].find(segs[j]) for j in range(n_tables)]
            elif op == "foi":
                if any(orcs[j].size() + len(np.unique(segs[j])) > tabs[j].capacity - 32 for j in range(n_tables)):
                    continue
                out, found = grp.find_or_insert(T(keys, dev), d_off)
                exp = [orcs[j].find_or_insert(segs[j]) for j in range(n_tables)]
            else:
                step += 1
                g = (rng.standard_normal((keys.size, dim)) * 0.05).astype(np.float32)
                if opt == "adagrad":
                    grp.apply_adagrad(T(keys, dev), d_off, T(g, dev), lr=0.05)
                else:
                    grp.apply_adam(T(keys, dev), d_off, T(g, dev), lr=0.01, step=step)
                for j in range(n_tables):
                    gj = g[off[j]:off[j + 1]]
                    if segs[j].size:
                        orcs[j].apply_adagrad(segs[j], gj, 0.05, 1e-10) if opt == "adagrad" else orcs[j].apply_adam(segs[j], gj, 0.01, 0.9, 0.999, 1e-8, step)
                continue
            assert np.array_equal(found.cpu().numpy(), np.concatenate([e[1] for e in exp]))
            np.testing.assert_allclose(out.cpu().numpy(), np.concatenate([e[0] for e in exp]), rtol=RTOL, atol=ATOL)
        elif op in ("pooled", "papply"):
            bpt, lens, off, segs, keys = batch(True)
            if keys.size == 0:
                continue
            mode = "sum" if rng.random() < 0.5 else "mean"
            located = torch.empty(keys.size, dtype=torch.int64, device=dev) if rng.random() < 0.5 else None
            out, found = grp.find_pooled(T(keys, dev), T(off, dev), mode, located=located)
            exp_rows = np.concatenate([orcs[j].find(segs[j])[0] for j in range(n_tables)])
            np.testing.assert_allclose(out.cpu().numpy(), oracle.pool_rows(exp_rows, off, mode), rtol=RTOL, atol=ATOL)
            if op == "papply":
                step += 1
                bg = (rng.standard_normal((lens.size, dim)) * 0.05).astype(np.float32)
                bag_of = np.repeat(np.arange(lens.size), lens).astype(np.int64)
                kwargs = dict(lr=0.05) if opt == "adagrad" else dict(lr=0.01, step=step)
                grp.apply_pooled(T(keys, dev), T(off, dev), T(bg, dev), T(bag_of, dev), opt, located=located, **kwargs)
                for j in range(n_tables):
                    lo, hi = off[j * bpt], off[(j + 1) * bpt]
                    if hi > lo:
                        gj = bg[bag_of[lo:hi]]
                        orcs[j].apply_adagrad(segs[j], gj, 0.05, 1e-10) if opt == "adagrad" else orcs[j].apply_adam(segs[j], gj, 0.01, 0.9, 0.999, 1e-8, step)
        else:
            j = int(rng.integers(0, n_tables))
            k = univ[j][rng.integers(0, 400, int(rng.integers(1, 200)))]
            if op == "insert":
                if orcs[j].size() + len(np.unique(k)) > tabs[j].capacity - 32:
                    continue
                rows = rng.standard_normal((k.size, dim)).astype(np.float32)
                tabs[j].insert(T(k, dev), T(rows, dev)); orcs[j].insert(k, rows)
            elif op == "remove":
                assert np.array_equal(tabs[j].remove(T(k, dev)).cpu().numpy(), orcs[j].remove(k))
            else:
                tabs[j].reserve(int(rng.integers(orcs[j].size() + 64, 4000)))   # planes move: the group re-reads them
    for j in range(n_tables):
        ga = [x.cpu().numpy() for x in tabs[j].export(with_state=True) if x is not None]
        oa = [x for x in orcs[j].export(with_state=True) if x is not None]
        ia, io = np.argsort(ga[0]), np.argsort(oa[0])
        assert np.array_equal(ga[0][ia], oa[0][io]), j
        for x, z in zip(ga[1:], oa[1:]):
            np.testing.assert_allclose(x[ia], z[io], rtol=RTOL, atol=ATOL)
    grp.close()


def test_dedup_keys_and_padded_partition(dev):
    """The sync-free pieces of a de-duplicated sharded lookup: mee_dedup_keys (distinct keys + EMPTY padding + inverse) and
    mee_partition_padded (padding belongs to no shard) against the oracle's dedup / partition of the same keys."""
    rng = np.random.default_rng(12)
    t = LookupTable(1024, 16, device=dev, max_batch=8192)
    u = synth.keys_np(70, 0, 900)
    keys = u[np.minimum(rng.zipf(1.2, 5000) - 1, 899)].copy()
    keys[7] = oracle.EMPTY_KEY; keys[11] = oracle.RECLAIMED_KEY
    uniq, inverse = t.dedup_keys(T(keys, dev), miss_index=8192)
    uniq, inverse = uniq.cpu().numpy(), inverse.cpu().numpy()
    ou = oracle.dedup_sum(keys, None, 16)[0]
    nu = ou.size
    at = np.flatnonzero(uniq != oracle.EMPTY_KEY)   # every distinct key once, EMPTY everywhere else (padding may lie between the keys)
    assert at.size == nu and np.array_equal(np.sort(uniq[at]), np.sort(ou))
    valid = (keys != oracle.EMPTY_KEY) & (keys != oracle.RECLAIMED_KEY)
    assert np.array_equal(uniq[inverse[valid]], keys[valid]) and (inverse[~valid] == 8192).all()
    assert t.size() == 0 and t.status() == STATUS_RESERVED_KEY      # scratch only; the tombstone value was flagged
    for g in (1, 3, 8):
        r = Router(g, 8192, device=dev)
        send, counts, perm = r.partition(T(uniq, dev), skip_padding=True)
        send, counts, perm = send.cpu().numpy(), counts.cpu().numpy(), perm.cpu().numpy()
        es, ec, ep = oracle.partition(uniq[at], g)              # the oracle partitions the keys without the padding: same order, positions through `at`
        assert np.array_equal(counts, ec) and counts.sum() == nu
        assert np.array_equal(send[:nu], es) and np.array_equal(perm[:nu], at[ep])


def test_new_entry_points_accept_empty_batches(dev):
    """n = 0 through every grouped / pooled / dedup entry point: a no-op, not an error."""
    from meepoembedding_amd import TableGroup
    dim = 64
    tabs = [LookupTable(256, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=1024) for _ in range(2)]
    grp = TableGroup(tabs, max_apply_batch=1024)
    e_keys = torch.zeros(0, dtype=torch.int64, device=dev)
    e_rows = torch.zeros((0, dim), device=dev)
    off = torch.zeros(3, dtype=torch.int64, device=dev)
    assert grp.find(e_keys, off)[0].shape == (0, dim)
    assert grp.find_or_insert(e_keys, off)[0].shape == (0, dim)
    grp.apply_adagrad(e_keys, off, e_rows, lr=0.1)
    out, _ = grp.find_pooled(e_keys, torch.zeros(5, dtype=torch.int64, device=dev))      # 2 bags per table, all empty
    assert out.shape == (4, dim) and not bool(out.any())
    grp.apply_pooled(e_keys, torch.zeros(5, dtype=torch.int64, device=dev), torch.zeros((4, dim), device=dev), e_keys, "adagrad", lr=0.1)
    u, inv = tabs[0].dedup_keys(e_keys)
    assert u.numel() == 0 and inv.numel() == 0
    s, c, p = Router(4, 64, device=dev).partition(e_keys, skip_padding=True)
    assert c.tolist() == [0, 0, 0, 0]
    tabs[0].apply_adagrad(e_keys, torch.zeros((3, dim), device=dev), lr=0.1, grad_index=e_keys)
    assert all(t.size() == 0 and t.status() == 0 for t in tabs)

"""mee_dedup_sum (round 5: sync-free, on the bucketed machinery) and the two round-4 advisor findings on the same machinery, against the C oracle:
the padded outputs on uniform / skewed / single-key batches over several steps of one stream (so that hot keys get buckets of their own and
the window path runs), buckets beyond the LDS list, hash prefixes beyond 32 bits (crafted keys), and the "apply_bucket_max" knob at max_batch.
Bar: integer outputs (distinct keys, counts, inverse) bit-exact; summed rows <= 1e-6 relative (fp64 sums rounded once: SPEC.md §4)."""
import zlib

import numpy as np
import pytest
import torch

import oracle
from meepoembedding_amd import OPT_ADAGRAD, LookupTable, synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-9
M64 = (1 << 64) - 1


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check_padded(keys, grads, dim, uniq, gs, cnt, inv, miss, what):
    """padded outputs of mee_dedup_sum == the oracle's compact ones"""
    ou, ogs, _, ocnt = oracle.dedup_sum(keys, grads if grads is not None else np.zeros((keys.size, dim), np.float32), dim)
    at = np.flatnonzero(uniq != oracle.EMPTY_KEY)
    assert at.size == ou.size, f"{what}: {at.size} distinct keys, the oracle has {ou.size}"
    a, b = at[np.argsort(uniq[at])], np.argsort(ou)
    assert np.array_equal(uniq[a], ou[b]), f"{what}: distinct keys"
    assert np.array_equal(cnt[a].astype(np.int64), ocnt[b].astype(np.int64)), f"{what}: counts"
    pad = np.ones(uniq.size, bool); pad[at] = False
    assert (cnt[pad] == 0).all(), f"{what}: padding must carry count 0"
    valid = keys > oracle.EMPTY_KEY + 1
    assert np.array_equal(uniq[inv[valid]], keys[valid]) and (inv[~valid] == miss).all(), f"{what}: inverse"
    if grads is not None:
        np.testing.assert_allclose(gs[a], ogs[b], rtol=RTOL, atol=ATOL, err_msg=what)
        once = a[cnt[a] == 1]                                   # a key that occurs once: its row, bit for bit
        src = np.full(uniq.size, -1, np.int64); src[inv[valid]] = np.flatnonzero(valid)
        assert np.array_equal(gs[once], grads[src[once]]), f"{what}: single occurrences must be copies"


def _stream(kind, step, n, rng, pool):
    if kind == "uniform":
        return pool[rng.integers(0, pool.size, n)]
    if kind == "zipf":
        return pool[(rng.zipf(1.05, n) - 1) % pool.size]
    if kind == "one_key":
        return np.full(n, pool[7], np.int64)
    if kind == "hot_mix":   # a few VERY hot keys (several windows each), ~150 warm ones (wave- and block-level runs), a uniform tail; the hot set rotates
        hot = pool[(np.arange(150) + 31 * step) % 400]
        reps = rng.integers(9, 700, size=150); reps[:3] = (n // 5, 5000, 1100)
        k = np.concatenate([np.repeat(hot, reps), pool[400 + rng.integers(0, pool.size - 400, n)]])[:n]
        rng.shuffle(k)
        return k
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,dim", [("uniform", 5000, 64), ("uniform", 300_000, 64), ("zipf", 300_000, 64), ("zipf", 120_000, 128),
                                         ("hot_mix", 300_000, 16), ("hot_mix", 200_000, 40), ("one_key", 70_000, 64), ("zipf", 1_000_000, 16)])
def test_dedup_sum_padded_over_a_stream(dev, kind, n, dim):
    rng = np.random.default_rng(zlib.crc32(f"{kind}/{n}/{dim}".encode()))
    pool = synth.keys_np(811, 0, max(2000, n // 2))
    t = LookupTable(64, dim, device=dev, max_batch=n)            # scratch only: no table row is read
    for step in range(4):                                        # one stream: from the second batch on the hot keys have buckets (windows) of their own
        keys = _stream(kind, step, n, rng, pool).copy()
        keys[min(123, n - 1)] = oracle.EMPTY_KEY
        keys[min(977, n - 1)] = oracle.EMPTY_KEY + 1              # a tombstone value in a batch: skipped, reported
        grads = rng.standard_normal((n, dim)).astype(np.float32)
        uniq, gs, cnt, inv = (x.cpu().numpy() for x in t.dedup_sum(T(keys, dev), T(grads, dev), miss_index=-5))
        _check_padded(keys, grads, dim, uniq, gs, cnt, inv, -5, f"{kind} step {step}")
        if step == 1:                                            # keys, counts and inverse only
            u2, g2, c2, i2 = t.dedup_sum(T(keys, dev), None, miss_index=n)
            assert g2 is None
            _check_padded(keys, None, dim, u2.cpu().numpy(), None, c2.cpu().numpy(), i2.cpu().numpy(), n, f"{kind} keys only")
    assert t.status() & ~2 == 0                                   # (bit 1: the tombstone value the batches carried)
    # compact form == the oracle's, and the other dedup of the same machinery still agrees with it on the same stream
    keys = _stream(kind, 9, n, rng, pool)
    grads = rng.standard_normal((n, dim)).astype(np.float32)
    cu, cg, cc, ci = (x.cpu().numpy() for x in t.dedup_sum(T(keys, dev), T(grads, dev), compact=True))
    ou, ogs, _, ocnt = oracle.dedup_sum(keys, grads, dim)
    a, b = np.argsort(cu), np.argsort(ou)
    assert np.array_equal(cu[a], ou[b]) and np.array_equal(cc[a], ocnt[b]) and np.array_equal(cu[ci], keys)
    np.testing.assert_allclose(cg[a], ogs[b], rtol=RTOL, atol=ATOL)
    uk, ik = (x.cpu().numpy() for x in t.dedup_keys(T(keys, dev)))
    assert np.array_equal(np.sort(uk[uk != oracle.EMPTY_KEY]), ou[b]) and np.array_equal(uk[ik], keys)


def test_raw_stream_operators_keep_their_skew_state_across_applies(dev):
    """A training step that aggregates before its exchange runs dedup_keys + dedup_sum over the RAW batch and the apply over its DISTINCT keys on ONE table.  The apply's
    "no skew" report must not send the next step's dedups back to the plan without buckets for hot keys (round 5's first form: dedup_sum 485 us average, 2.5 ms worst, per
    1M-key Zipf batch in that loop).  Results against the oracle every step; time: the dedup_sum of the interleaved loop within 2.5x of the same operator alone on the same stream."""
    n, dim, n_keys = 400_000, 64, 2_000_000
    rng = np.random.default_rng(77)
    pool = synth.keys_np(813, 0, n_keys)
    t = LookupTable(int(n_keys / 0.7), dim, device=dev, max_batch=n, optimizer=OPT_ADAGRAD)
    for s0 in range(0, n_keys, 400_000):
        kk = pool[s0:s0 + 400_000]
        t.insert(T(kk, dev), T(synth.rows_np(kk, dim, 3), dev))
    batches = [pool[(rng.zipf(1.05, n) - 1) % n_keys] for _ in range(4)]
    grads = rng.standard_normal((n, dim)).astype(np.float32)
    g = T(grads, dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_sum(k):
        e0.record(); out = t.dedup_sum(k, g); e1.record(); torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) * 1e3

    alone = []
    for i in range(8):                                            # the operator alone on the stream: its steady state
        _, us = timed_sum(T(batches[i % 4], dev))
        alone.append(us)
    loop = []
    for i in range(8):                                            # the training step's order of operators on one table
        k = T(batches[i % 4], dev)
        uk, ik = t.dedup_keys(k)
        (uniq, gs, cnt, inv), us = timed_sum(k)
        loop.append(us)
        t.apply_adagrad(uniq, gs, lr=0.01)                        # the padded distinct keys (EMPTY holes are skipped) and their summed rows
        if i >= 6:
            _check_padded(batches[i % 4], grads, dim, uniq.cpu().numpy(), gs.cpu().numpy(), cnt.cpu().numpy(), inv.cpu().numpy(), -1, f"interleaved step {i}")
            assert np.array_equal(uk.cpu().numpy()[ik.cpu().numpy()], batches[i % 4])
    assert t.status() == 0
    a, b = float(np.median(alone[3:])), float(np.median(loop[3:]))
    print(f"dedup_sum per {n} Zipf keys: alone {a:.0f} us, between dedup_keys and an apply over the distinct keys {b:.0f} us (worst {max(loop[3:]):.0f})")
    assert b <= 2.5 * a + 20.0 and max(loop[3:]) <= 4.0 * a + 40.0


def test_dedup_sum_is_graph_capturable(dev):
    """sync-free means capturable: the same launches replayed on new contents of the same buffers"""
    n, dim = 50_000, 64
    rng = np.random.default_rng(5)
    pool = synth.keys_np(812, 0, 9000)
    t = LookupTable(64, dim, device=dev, max_batch=n)
    k = torch.empty(n, dtype=torch.int64, device=dev); g = torch.empty((n, dim), device=dev)
    k.copy_(T(pool[rng.integers(0, pool.size, n)], dev)); g.normal_()
    t.dedup_sum(k, g)                                             # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = t.dedup_sum(k, g, miss_index=-1)
    for step in range(3):
        keys = pool[(rng.zipf(1.05, n) - 1) % pool.size] if step else pool[rng.integers(0, pool.size, n)]
        grads = rng.standard_normal((n, dim)).astype(np.float32)
        k.copy_(T(keys, dev)); g.copy_(T(grads, dev))
        graph.replay()
        torch.cuda.synchronize()
        uniq, gs, cnt, inv = (x.cpu().numpy() for x in outs)
        _check_padded(keys, grads, dim, uniq, gs, cnt, inv, -1, f"replay {step}")


def _inv_mix64b(y):
    """inverse of SPEC.md §1's mix64b (murmur3 fmix64): x ^= x >> 33 is its own inverse, the multipliers are odd"""
    i1, i2 = pow(0xFF51AFD7ED558CCD, -1, 1 << 64), pow(0xC4CEB9FE1A85EC53, -1, 1 << 64)
    y ^= y >> 33; y = (y * i2) & M64
    y ^= y >> 33; y = (y * i1) & M64
    y ^= y >> 33
    return y


def _colliding_keys(count, n_buckets, bucket, low32):
    """`count` distinct keys whose mix64b values share their low 32 bits AND that the partition of a small batch (n_buckets = ceil(n / 128)) sends to
    one bucket: what round 4's 32-bit pass prefixes could not separate."""
    from oracle import pyspec
    out, hi = [], 0
    while len(out) < count:
        key = _inv_mix64b((hi << 32) | low32)
        hi += 1
        if key >= 1 << 63:
            key -= 1 << 64
        if key <= oracle.EMPTY_KEY + 1:
            continue
        assert pyspec.mix64b(key & M64) & 0xFFFFFFFF == low32
        if (pyspec.mix64(key & M64) * n_buckets) >> 64 == bucket:
            out.append(key)
    return np.array(out, np.int64)


def test_dedup_passes_split_beyond_32_hash_bits(dev):
    """1500 distinct keys of ONE bucket whose mix64b values agree in their low 32 bits: the pass driver has to split on bits 32 and up (it used to stop
    at 32 and drop the pass silently: stale inverse entries, missing keys).  dedup_keys, dedup_sum and assign on the same batch."""
    dim, n = 16, 2048
    n_buckets = (n + 127) // 128
    crafted = _colliding_keys(1500, n_buckets, 3, 0x5EED5EED)
    rng = np.random.default_rng(3)
    keys = np.concatenate([crafted, crafted[:300], synth.keys_np(813, 0, n - 1800)])
    rng.shuffle(keys)
    t = LookupTable(8192, dim, device=dev, max_batch=n)
    o = oracle.OracleTable(8192, dim)
    base = np.unique(keys)[::2]
    rows0 = synth.rows_np(base, dim, 2)
    t.insert(T(base, dev), T(rows0, dev)); o.insert(base, rows0)
    for rep in range(2):
        uk, ik = (x.cpu().numpy() for x in t.dedup_keys(T(keys, dev)))
        assert np.array_equal(np.sort(uk[uk != oracle.EMPTY_KEY]), np.unique(keys)) and np.array_equal(uk[ik], keys)
        grads = rng.standard_normal((n, dim)).astype(np.float32)
        uniq, gs, cnt, inv = (x.cpu().numpy() for x in t.dedup_sum(T(keys, dev), T(grads, dev)))
        _check_padded(keys, grads, dim, uniq, gs, cnt, inv, -1, f"crafted rep {rep}")
        v = rng.standard_normal((n, dim)).astype(np.float32)
        f = t.assign(T(keys, dev), T(v, dev)).cpu().numpy()
        assert np.array_equal(f.astype(bool), np.asarray(o.assign(keys, v)).astype(bool))
    out, found = t.find(T(base, dev))
    eo, ef = o.find(base)
    assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo) and t.status() == 0


@pytest.mark.parametrize("bucket_max", [32, 64, 128, 352])
def test_apply_bucket_max_knob_at_max_batch(dev, bucket_max):
    """the scratch of the bucketed machinery is strided for the bucket count of the DEFAULT bucket size at max_batch; a smaller
    "apply_bucket_max" asked for up to 7680 buckets on strides of 3200 — out-of-bounds device writes from a knob documented as 'never changes
    results'.  The count is clamped now: apply, located apply behind the forward, dedup_keys, dedup_sum and assign at n = max_batch, every value."""
    dim, n, n_keys = 16, 1_000_000, 400_000
    rng = np.random.default_rng(bucket_max)
    keys = synth.keys_np(814, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    t = LookupTable(1 << 20, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=n, initial_accumulator=0.1)
    o = oracle.OracleTable(1 << 20, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    t.insert(T(keys, dev), T(rows, dev)); o.insert(keys, rows)
    t.set_tuning("apply_bucket_max", bucket_max)
    bk = keys[rng.integers(0, n_keys, n)]
    bk[: n // 50] = keys[11]                                       # one hot key: split buckets / windows as well
    rng.shuffle(bk)
    bkt = T(bk, dev)
    for step in range(2):
        g = (rng.standard_normal((n, dim)) * 0.01).astype(np.float32)
        if step == 0:
            t.apply_adagrad(bkt, T(g, dev), lr=0.05)
        else:
            _, _, slots = t.find_located(bkt, prepare_apply=True)
            t.apply_adagrad(bkt, T(g, dev), lr=0.05, slots=slots)
        o.apply_adagrad(bk, g, 0.05, 1e-10)
        uk, ik = (x.cpu().numpy() for x in t.dedup_keys(bkt))
        assert np.array_equal(uk[ik], bk)
        uniq, gs, cnt, inv = (x.cpu().numpy() for x in t.dedup_sum(bkt, T(g, dev)))
        _check_padded(bk, g, dim, uniq, gs, cnt, inv, -1, f"bucket_max {bucket_max} step {step}")
        v = rng.standard_normal((n, dim)).astype(np.float32)
        f = t.assign(bkt, T(v, dev)).cpu().numpy()
        assert np.array_equal(f.astype(bool), np.asarray(o.assign(bk, v)).astype(bool))
    assert t.status() == 0
    e = [x.cpu().numpy() for x in t.export(with_state=True)[:3]]
    eo = o.export(with_state=True)[:3]
    a, b = np.argsort(e[0]), np.argsort(eo[0])
    assert np.array_equal(e[0][a], eo[0][b])
    np.testing.assert_allclose(e[1][a], eo[1][b], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(e[2][a], eo[2][b], rtol=RTOL, atol=ATOL)

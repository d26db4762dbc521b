"""Hot/cold tier (meepoembedding_amd/tiered.py): the pair must behave like ONE table.  CPU: the host logic over two
oracle-backed tiers vs a single oracle table.  GPU: hot tier in HBM + cold tier with rows in pinned host DRAM (the HIP
kernels read/write them over PCIe) vs the same single oracle table."""
import numpy as np
import pytest
import torch

import oracle
from meepoembedding_amd import synth
from meepoembedding_amd.tiered import TieredLookupTable

DIM = 16
KW = dict(default_value=0.5, initial_accumulator=0.1, initializer=oracle.INIT_UNIFORM, init_scale=0.05, init_seed=9)


def _sorted(exp):
    keys = exp[0]
    order = np.argsort(keys)
    return [None if x is None else x[order] for x in exp]


def _drive(tiered, ref, dev, opt, n_iter=40, seed=0, DIM=DIM):
    rng = np.random.default_rng(seed)
    universe = synth.keys_np(70 + seed, 0, 3000)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    step = 0
    for it in range(n_iter):
        n = int(rng.integers(1, 900))
        idx = np.minimum(rng.zipf(1.3, size=n) - 1, universe.size - 1) if rng.random() < 0.5 else rng.integers(0, universe.size, n)
        keys = universe[idx]
        rows = rng.standard_normal((n, DIM)).astype(np.float32)
        op = rng.choice(["insert", "insert", "assign", "remove", "find", "find", "find_or_insert", "apply", "promote", "demote", "rebalance"])
        if op == "insert":
            tiered.insert(T(keys), T(rows)); ref.insert(keys, rows)
        elif op == "assign":
            assert np.array_equal(tiered.assign(T(keys), T(rows)).cpu().numpy(), ref.assign(keys, rows))
        elif op == "remove":
            keys = keys[: max(1, n // 5)]
            assert np.array_equal(tiered.remove(T(keys)).cpu().numpy(), ref.remove(keys))
        elif op == "find":
            out, found = tiered.find(T(keys)); eo, ef = ref.find(keys)
            assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
        elif op == "find_or_insert":
            out, found = tiered.find_or_insert(T(keys)); eo, ef = ref.find_or_insert(keys)
            assert np.array_equal(found.cpu().numpy(), ef) and np.array_equal(out.cpu().numpy(), eo)
        elif op == "apply":
            step += 1
            g = (rows * 0.01).astype(np.float32)
            if opt == oracle.OPT_ADAGRAD:
                tiered.apply_adagrad(T(keys), T(g), lr=0.02, eps=1e-10); ref.apply_adagrad(keys, g, 0.02, 1e-10)
            else:
                tiered.apply_adam(T(keys), T(g), lr=0.002, step=step); ref.apply_adam(keys, g, 0.002, 0.9, 0.999, 1e-8, step)
        elif op == "rebalance":
            if tiered.policy:
                tiered.rebalance(max_moves=300)   # policy-driven migration must not change anything observable either
        elif op == "promote":
            # migration must not change anything observable — neither through zero-copy reads nor through the staged transfer
            # (host-side gather + asynchronous copy on a side stream; real tables only)
            if it % 2 and hasattr(tiered.cold, "plane_host_view"):
                tiered.promote(T(keys[:200]), staged=True)
            else:
                tiered.promote(T(keys[:200]))
        else:
            tiered.demote(T(keys[:200]))
        assert tiered.size() == ref.size(), (it, op)
    got = _sorted([None if x is None else x.cpu().numpy() for x in tiered.export(with_state=True)])
    exp = _sorted(list(ref.export(with_state=True)))
    assert np.array_equal(got[0], exp[0])
    for a, b in zip(got[1:], exp[1:]):
        if b is not None:
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)
    return tiered


@pytest.mark.parametrize("opt", [oracle.OPT_ADAGRAD, oracle.OPT_ADAM])
def test_tiered_cpu_logic(built, opt):
    from _cpu_backend import CpuTable
    hot = CpuTable(1024, DIM, optimizer=opt, track_hits=True, **KW)
    cold = CpuTable(8192, DIM, optimizer=opt, track_hits=True, **KW)
    ref = oracle.OracleTable(16384, DIM, optimizer=opt, **KW)
    t = _drive(TieredLookupTable(hot, cold, hot_key_limit=600, sample_every=2), ref, torch.device("cpu"), opt, seed=int(opt))
    assert 0 < hot.size() <= 600 and cold.size() > 0, "both tiers must have been used"


# dims: 16 = configs[0], 64 = configs[4] (the hot/cold tier's stated shape), 128 = configs[3]
@pytest.mark.gpu
@pytest.mark.parametrize("dim", [16, 64, 128])
@pytest.mark.parametrize("opt", [oracle.OPT_ADAGRAD, oracle.OPT_ADAM])
def test_tiered_hbm_plus_pinned_host(dev, opt, dim):
    from meepoembedding_amd import LookupTable, _lib
    hot = LookupTable(1024, dim, device=dev, optimizer=opt, max_batch=4096, track_hits=True, **KW)
    cold = LookupTable(8192, dim, device=dev, optimizer=opt, max_batch=4096, value_memory=_lib.MEM_HOST_PINNED, track_hits=True, **KW)
    ref = oracle.OracleTable(16384, dim, optimizer=opt, **KW)
    _drive(TieredLookupTable(hot, cold, hot_key_limit=600, sample_every=2), ref, dev, opt, seed=10 + int(opt), DIM=dim)
    assert 0 < hot.size() <= 600 and cold.size() > 0


@pytest.mark.gpu
def test_policy_moves_the_hot_set_into_hbm(dev):
    """Skewed lookups over keys that all start cold: after a few observe + rebalance rounds the frequently used keys
    sit in the HBM tier and the cold share of the lookups has collapsed; nothing observable changed."""
    from meepoembedding_amd import LookupTable, _lib
    n_keys, hot_limit = 20000, 2000
    hot = LookupTable(4096, DIM, device=dev, max_batch=8192, track_hits=True)
    cold = LookupTable(32768, DIM, device=dev, max_batch=32768, value_memory=_lib.MEM_HOST_PINNED, track_hits=True)
    keys = synth.keys_np(33, 0, n_keys); rows = synth.rows_np(keys, DIM, 2)
    cold.insert(torch.from_numpy(keys).to(dev), torch.from_numpy(rows).to(dev))
    t = TieredLookupTable(hot, cold, hot_key_limit=hot_limit, sample_every=2, promote_threshold=2)
    rng = np.random.default_rng(5)

    def batch():
        return keys[np.minimum(rng.zipf(1.2, size=8000) - 1, n_keys - 1)]

    def cold_share(b):
        _, f = hot.find(torch.from_numpy(b).to(dev))
        return 1.0 - float(f.float().mean())

    first = cold_share(batch())
    for rnd in range(4):
        for _ in range(6):
            b = batch()
            out, found = t.find(torch.from_numpy(b).to(dev))
            assert bool(found.all()) and np.array_equal(out.cpu().numpy(), synth.rows_np(b, DIM, 2))
        t.rebalance(max_moves=4000)
    last = cold_share(batch())
    assert first > 0.99 and last < 0.35, (first, last)
    assert hot.size() <= hot_limit and hot.size() + cold.size() == n_keys


@pytest.mark.gpu
def test_training_loop_rebalances_itself(dev):
    """rebalance_every = K: the pair runs its placement policy behind every K-th optimizer step of a find + sparse-Adagrad loop — keys move between
    the tiers WITH their accumulators while the loop trains them —, the cold share of the lookups collapses, and the trained pair equals ONE oracle
    table that saw the same steps (SPEC.md §3-§4: placement is not observable)."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, _lib
    n_keys, hot_limit, dim = 20000, 2500, 16
    hot = LookupTable(4096, dim, device=dev, max_batch=8192, track_hits=True, optimizer=OPT_ADAGRAD, initial_accumulator=0.1)
    cold = LookupTable(32768, dim, device=dev, max_batch=32768, value_memory=_lib.MEM_HOST_PINNED, track_hits=True, optimizer=OPT_ADAGRAD, initial_accumulator=0.1)
    o = oracle.OracleTable(65536, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    keys = synth.keys_np(34, 0, n_keys); rows = synth.rows_np(keys, dim, 2)
    cold.insert(torch.from_numpy(keys).to(dev), torch.from_numpy(rows).to(dev)); o.insert(keys, rows)
    t = TieredLookupTable(hot, cold, hot_key_limit=hot_limit, sample_every=2, promote_threshold=2, rebalance_every=5, rebalance_max_moves=4000)
    rng = np.random.default_rng(6)
    shares = []
    for step in range(30):
        b = keys[np.minimum(rng.zipf(1.2, size=6000) - 1, n_keys - 1)]
        g = (rng.standard_normal((b.size, dim)) * 0.01).astype(np.float32)
        bt = torch.from_numpy(b).to(dev)
        if step % 5 == 0:
            shares.append(1.0 - float(hot.find(bt)[1].float().mean()))
        out, found = t.find(bt)
        eo, ef = o.find(b)
        assert bool(found.all()) and np.array_equal(found.cpu().numpy(), ef)
        np.testing.assert_allclose(out.cpu().numpy(), eo, rtol=1e-6, atol=1e-9)
        t.apply_adagrad(bt, torch.from_numpy(g).to(dev), lr=0.05)
        o.apply_adagrad(b, g, 0.05, 1e-10)
    assert len(t.rebalance_log) == 6 and sum(p for _, p, _ in t.rebalance_log) > 0
    assert shares[0] > 0.99 and shares[-1] < 0.35, shares
    assert hot.size() <= hot_limit and hot.size() + cold.size() == n_keys and hot.status() == 0 and cold.status() == 0
    eh, ec = hot.export(with_state=True), cold.export(with_state=True)   # (ONE export per table: the order of an export's pairs is unspecified, call by call)
    ek, ev, ea = (torch.cat([eh[i], ec[i]]).cpu().numpy() for i in range(3))
    ok, ov, oa, _ = o.export(with_state=True)
    a, b_ = np.argsort(ek), np.argsort(ok)
    assert np.array_equal(ek[a], ok[b_])
    np.testing.assert_allclose(ev[a], ov[b_], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ea[a], oa[b_], rtol=1e-6, atol=1e-9)

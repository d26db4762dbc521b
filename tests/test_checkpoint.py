"""Checkpoint format + ranged export + eviction.  CPU: the file format and the re-sharding filter
on oracle-backed shards.  GPU: mee_export_range against the oracle's export, save -> load into a table of another
capacity (rows and optimizer state bit-identical), eviction by hit counters."""
import json
import os
import types

import numpy as np
import pytest
import torch

import oracle
from meepoembedding_amd import checkpoint, synth

DIM = 8


def _fill(t, n, seed, to=lambda x: x):
    rng = np.random.default_rng(seed)
    keys = synth.keys_np(seed, 0, n)
    t.insert(to(torch.from_numpy(keys)), to(torch.from_numpy(rng.standard_normal((n, DIM)).astype(np.float32))))
    t.apply_adam(to(torch.from_numpy(keys[: n // 2])), to(torch.from_numpy(rng.standard_normal((n // 2, DIM)).astype(np.float32))), lr=0.01, step=1)
    return keys


def _sorted(exp):
    k = exp[0].cpu()
    i = torch.argsort(k)
    return [k[i]] + [x.cpu()[i] for x in exp[1:] if x is not None]


def test_checkpoint_format_and_resharding_cpu(built, tmp_path):
    from _cpu_backend import CpuRouter, CpuTable
    mk = lambda: CpuTable(8192, DIM, optimizer=oracle.OPT_ADAM)
    src = mk()
    keys = _fill(src, 3000, 5)
    p = str(tmp_path / "one")
    assert checkpoint.save_table(src, p) == 3000
    meta = json.load(open(os.path.join(p, "meta.json")))
    assert meta == {"format": "meepo-table-v1", "dim": DIM, "optimizer": oracle.OPT_ADAM, "n": 3000, "planes": ["values", "state1", "state2"], "extra": {}}
    assert os.path.getsize(os.path.join(p, "keys.i64")) == 3000 * 8 and os.path.getsize(os.path.join(p, "state2.f32")) == 3000 * DIM * 4
    dst = mk()
    assert checkpoint.load_into(dst, p, chunk_pairs=700) == 3000
    for a, b in zip(_sorted(src.export(with_state=True)), _sorted(dst.export(with_state=True))):
        assert torch.equal(a, b)
    # a 2-way checkpoint loaded by a 3-way job: every pair lands on its new owner, nothing lost, nothing doubled
    root = str(tmp_path / "sharded")
    two = [types.SimpleNamespace(local=mk(), rank=r, world=2) for r in range(2)]
    own2 = oracle.hash_batch(keys, 1, 2)[2]
    ek, ev, e1, e2 = src.export(with_state=True)
    for r, sh in enumerate(two):
        m = torch.from_numpy(oracle.hash_batch(ek.numpy(), 1, 2)[2] == r)
        sh.local.import_(ek[m], ev[m], e1[m], e2[m])
        checkpoint.save_sharded(sh, root)
    assert sum(s.local.size() for s in two) == 3000 and int((own2 == 0).sum()) == two[0].local.size()
    three = [types.SimpleNamespace(local=mk(), rank=r, world=3) for r in range(3)]
    loaded = [checkpoint.load_sharded(sh, root, CpuRouter(3).owner, chunk_pairs=512) for sh in three]
    assert sum(loaded) == 3000
    merged = [torch.cat(x) for x in zip(*[[y for y in sh.local.export(with_state=True)] for sh in three])]
    for r, sh in enumerate(three):
        assert (oracle.hash_batch(sh.local.export()[0].numpy(), 1, 3)[2] == r).all()
    for a, b in zip(_sorted(src.export(with_state=True)), _sorted(merged)):
        assert torch.equal(a, b)
    # same world size: a rank reads only its own directory
    again = types.SimpleNamespace(local=mk(), rank=1, world=2)
    assert checkpoint.load_sharded(again, root, CpuRouter(2).owner) == two[1].local.size()
    # an unfinished save (no meta.json) is refused; a wrong dim is refused
    os.remove(os.path.join(p, "meta.json"))
    with pytest.raises(FileNotFoundError):
        checkpoint.load_into(mk(), p)
    with pytest.raises(ValueError):
        checkpoint.load_into(CpuTable(64, DIM * 2), os.path.join(root, "shard-00000-of-00002"))


@pytest.mark.gpu
def test_export_range_and_checkpoint_gpu(dev, tmp_path):
    from meepoembedding_amd import OPT_ADAM, LookupTable
    t = LookupTable(20000, DIM, device=dev, optimizer=OPT_ADAM, max_batch=4096)
    o = oracle.OracleTable(20000, DIM, optimizer=oracle.OPT_ADAM)
    rng = np.random.default_rng(5)
    keys = synth.keys_np(5, 0, 12000)
    for s in range(0, 12000, 4096):
        k = keys[s:s + 4096]
        v = rng.standard_normal((k.size, DIM)).astype(np.float32)
        g = rng.standard_normal((k.size, DIM)).astype(np.float32)
        t.insert(torch.from_numpy(k).to(dev), torch.from_numpy(v).to(dev)); o.insert(k, v)
        t.apply_adam(torch.from_numpy(k).to(dev), torch.from_numpy(g).to(dev), lr=0.01, step=1); o.apply_adam(k, g, 0.01, 0.9, 0.999, 1e-8, 1)
    t.remove(torch.from_numpy(keys[::5]).to(dev)); o.remove(keys[::5])
    full = _sorted(t.export(with_state=True))
    # ranges that do not align with the kernel's 1024-slot spans, an empty range, a range past the end
    cuts = [0, 1, 777, 1024, 5000, 5000, 13001, t.capacity, t.capacity + 999]
    pieces = [t.export_range(a, b, with_state=True) for a, b in zip(cuts[:-1], cuts[1:])]
    assert pieces[4][0].numel() == 0
    merged = _sorted([torch.cat([p[i] for p in pieces]) for i in range(4)])
    for a, b in zip(full, merged):
        assert torch.equal(a, b)
    ok = o.export(with_state=True)
    srt = np.argsort(ok[0])
    assert np.array_equal(full[0].numpy(), ok[0][srt])
    for a, b in zip(full[1:], ok[1:]):
        np.testing.assert_allclose(a.numpy(), b[srt], rtol=1e-6, atol=1e-9)
    with pytest.raises(Exception):
        t.export_range(10, 5)
    # save in small ranges, load into a table of a different capacity, also through resized()
    p = str(tmp_path / "ckpt")
    assert t.save(p, chunk_slots=3000) == t.size()
    big = LookupTable(50000, DIM, device=dev, optimizer=OPT_ADAM, max_batch=1000)
    assert big.load(p) == t.size() and big.status() == 0
    for a, b in zip(full, _sorted(big.export(with_state=True))):
        assert torch.equal(a, b)
    for a, b in zip(full, _sorted(t.resized(30000, chunk=2048).export(with_state=True))):
        assert torch.equal(a, b)
    # rows only into a table with another optimizer: the state planes are not carried over
    from meepoembedding_amd import OPT_ADAGRAD
    other = LookupTable(20000, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.5)
    other.load(p)
    ek, ev, ea, _ = other.export(with_state=True)
    assert torch.equal(_sorted((ek, ev))[1], full[1]) and bool((ea == 0.5).all())


@pytest.mark.gpu
def test_evict_by_hits_gpu(dev):
    from meepoembedding_amd import LookupTable
    t = LookupTable(4096, DIM, device=dev, track_hits=True)
    keys = torch.from_numpy(synth.keys_np(8, 0, 2000)).to(dev)
    t.insert(keys, torch.ones(2000, DIM, device=dev))
    t.find_counted(keys[:500]); t.find_counted(keys[:100])
    assert t.evict(max_hits=0, limit=1200, reset=False) == 1200     # only never-hit keys, at most `limit`
    assert t.size() == 800 and bool(t.find(keys[:500])[1].all())
    assert t.evict(max_hits=1) == 300 + 400                        # the rest of the never-hit keys + the hit-once keys; counters reset
    assert t.size() == 100 and bool(t.find(keys[:100])[1].all())
    assert t.hits_scan(1, 1 << 30, 10).numel() == 0
    t.insert(keys[1000:1500], torch.ones(500, DIM, device=dev))     # tombstones are reused
    assert t.size() == 600 and t.status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_reserve_rehashes_in_place_gpu(dev, pinned):
    """mee_reserve: capacity changes, no observable does (rows, optimizer planes, hit counters, found masks)."""
    from meepoembedding_amd import OPT_ADAM, LookupTable, MeepoError, _lib
    t = LookupTable(4000, DIM, device=dev, optimizer=OPT_ADAM, max_batch=4096, track_hits=True,
                    value_memory=_lib.MEM_HOST_PINNED if pinned else _lib.MEM_HBM)
    o = oracle.OracleTable(1 << 16, DIM, optimizer=oracle.OPT_ADAM)
    rng = np.random.default_rng(9)
    keys = synth.keys_np(9, 0, 3000)
    v = rng.standard_normal((3000, DIM)).astype(np.float32)
    g = rng.standard_normal((3000, DIM)).astype(np.float32)
    to = lambda x: torch.from_numpy(x).to(dev)
    t.insert(to(keys), to(v)); o.insert(keys, v)
    t.apply_adam(to(keys[:2000]), to(g[:2000]), lr=0.01, step=1); o.apply_adam(keys[:2000], g[:2000], 0.01, 0.9, 0.999, 1e-8, 1)
    t.remove(to(keys[::7])); o.remove(keys[::7])
    t.find_counted(to(keys[1:200]))
    before = _sorted(t.export(with_state=True))
    hot_before = torch.sort(t.hits_scan(1, 1 << 30, 4096))[0]
    cap0 = t.capacity
    t.reserve(cap0)                                   # same bucket count: nothing to do
    assert t.capacity == cap0
    t.reserve(20000)
    assert t.capacity >= 20000 and t.capacity % 16 == 0 and t.status() == 0 and t.size() == o.size()
    for a, b in zip(before, _sorted(t.export(with_state=True))):
        assert torch.equal(a, b)
    assert torch.equal(hot_before, torch.sort(t.hits_scan(1, 1 << 30, 4096))[0])
    probe = np.concatenate([keys[:500], synth.keys_np(10, 0, 100)])
    out, found = t.find(to(probe)); eo, ef = o.find(probe)
    assert np.array_equal(found.cpu().numpy(), ef)
    np.testing.assert_allclose(out.cpu().numpy(), eo, rtol=1e-6, atol=1e-9)
    # the grown table takes more keys than the old capacity could hold, and keeps training
    more = synth.keys_np(11, 0, 12000)
    for s in range(0, 12000, 4096):
        mv = rng.standard_normal((more[s:s + 4096].size, DIM)).astype(np.float32)
        t.insert(to(more[s:s + 4096]), to(mv)); o.insert(more[s:s + 4096], mv)
    t.apply_adam(to(keys[:2000]), to(g[:2000]), lr=0.01, step=2); o.apply_adam(keys[:2000], g[:2000], 0.01, 0.9, 0.999, 1e-8, 2)
    assert t.status() == 0 and t.size() == o.size()
    ek = _sorted(t.export(with_state=True)); ok = o.export(with_state=True); srt = np.argsort(ok[0])
    assert np.array_equal(ek[0].numpy(), ok[0][srt])
    for a, b in zip(ek[1:], ok[1:]):
        np.testing.assert_allclose(a.numpy(), b[srt], rtol=1e-6, atol=1e-9)
    # shrinking: to fit is fine (tombstones are dropped on the way), below the stored count is refused and changes nothing
    n = t.size()
    with pytest.raises(MeepoError):
        t.reserve(n // 2)
    cap1 = t.capacity
    t.reserve(int(n / 0.8))
    assert t.capacity < cap1 and t.size() == n and t.status() == 0
    for a, b in zip(ek, _sorted(t.export(with_state=True))):
        assert torch.equal(a, b)
    assert t.maybe_grow(max_load=0.5) and t.capacity >= 2 * int(n / 0.8) and not t.maybe_grow(max_load=0.5)

"""Row-sharded path (SPEC.md §5).  CPU: world_size-2 gloo run of the host logic with oracle-backed shards, checked
against ONE global oracle table.  GPU: the same logic on the HIP backend through RCCL (world_size 1 on the 1-GPU
box: partition, all-to-all plumbing, local find, un-permute)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from meepoembedding_amd import synth
from meepoembedding_amd.sharded import ShardedLookupTable

DIM, NKEYS, BATCH = 16, 6000, 4000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches(world, dim=DIM):
    """Per-rank batches with cross-rank and in-batch duplicates."""
    rng = np.random.default_rng(100)
    keys = synth.keys_np(1, 0, NKEYS)
    out = []
    for r in range(world):
        idx = rng.integers(0, NKEYS, size=BATCH)
        out.append((keys[idx], rng.standard_normal((BATCH, dim)).astype(np.float32),
                    (rng.standard_normal((BATCH, dim)) * 0.01).astype(np.float32)))
    return out


def _global_reference(world, dim=DIM):
    """One table, batches applied in rank order == SPEC §5 'ordered by source rank then batch position'."""
    o = oracle.OracleTable(16384, dim, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
    b = _batches(world, dim)
    o.insert(np.concatenate([x[0] for x in b]), np.concatenate([x[1] for x in b]))
    o.apply_adagrad(np.concatenate([x[0] for x in b]), np.concatenate([x[2] for x in b]), 0.05, 1e-10)
    for r in range(world):
        o.remove(_removed(r))
    return o


def _removed(rank):
    return synth.keys_np(1, 0, NKEYS)[rank * 7::101]


def _run_rank(rank, world, port, backend, q, tiered=False, dim=DIM):
    try:
        _run_rank_body(rank, world, port, backend, q, tiered, dim)
    except BaseException as e:   # report at once: the parent must not sit out its queue timeout on the GPU box
        import traceback
        q.put(("error", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        raise


def _run_rank_body(rank, world, port, backend, q, tiered=False, DIM=DIM):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "gloo":
        from _cpu_backend import CpuRouter, CpuTable
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cpu")
        mk_local = lambda: CpuTable(16384, DIM, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1)
        if tiered:   # BASELINE configs[4]: every shard is a hot/cold pair
            from meepoembedding_amd.tiered import TieredLookupTable
            mk_flat = mk_local
            mk_local = lambda: TieredLookupTable(CpuTable(2048, DIM, optimizer=oracle.OPT_ADAGRAD, initial_accumulator=0.1), mk_flat(), hot_key_limit=1200)
        local = mk_local()
        router = CpuRouter(world)
    elif backend in ("gloo-gpu", "fake-rccl"):
        if backend == "fake-rccl":   # the exchange behind the C-ABI binds the shared-memory stand-in (tests/cabi/fake_rccl.cpp) instead of RCCL
            os.environ["MEE_RCCL_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libfake_rccl.so")
        # several ranks share ONE GPU; the exchange is staged through host memory over gloo — every HIP kernel of the
        # multi-rank path (partition with G > 1, find on received keys, un-permute) runs for real
        from meepoembedding_amd import OPT_ADAGRAD, LookupTable, Router
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mk_local = lambda: LookupTable(16384, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=world * BATCH)
        if tiered:   # configs[4] on the real backend: HBM table in front of a pinned-host-DRAM table, per shard
            from meepoembedding_amd import _lib
            from meepoembedding_amd.tiered import TieredLookupTable
            mk_local = lambda: TieredLookupTable(
                LookupTable(2048, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=world * BATCH),
                LookupTable(16384, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=world * BATCH,
                            value_memory=_lib.MEM_HOST_PINNED), hot_key_limit=1200)
        local = mk_local()
        router = Router(world, BATCH, device=dev)
    else:
        from meepoembedding_amd import OPT_ADAGRAD, LookupTable, Router
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        mk_local = lambda: LookupTable(16384, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=world * BATCH)
        local = mk_local()
        router = Router(world, BATCH, device=dev)
    try:
        sh = ShardedLookupTable(local, router)
        keys, rows, grads = (torch.from_numpy(x).to(dev) for x in _batches(world, DIM)[rank])
        sh.insert(keys, rows)
        # odd ranks aggregate their gradients before the exchange (one summed row per distinct key of the rank's batch travels), even ranks send every
        # occurrence: the owners' applies add both kinds up in fp64 — the global reference (un-aggregated, one table) must still be met within 1e-6
        sh.apply_adagrad(keys, grads, lr=0.05, eps=1e-10, dedup=bool(rank & 1))
        dist.barrier()
        sh.remove(torch.from_numpy(_removed(rank)).to(dev))
        dist.barrier()
        probe = torch.from_numpy(np.concatenate([synth.keys_np(1, 0, NKEYS)[rank::3], synth.keys_np(9, rank * 50, 50)])).to(dev)
        out, found = sh.find(probe)
        # the same lookup with pre-exchange dedup on a duplicate-heavy batch (+ a reserved key) must give the same answer
        dup = torch.cat([probe[:500].repeat(7), probe[-60:], torch.tensor([oracle.EMPTY_KEY], device=dev)])
        o1, f1 = sh.find(dup)
        if backend in ("gloo-gpu", "nccl") and not tiered:   # the all-to-all-free path over peer-mapped memory must agree bit for bit
            from meepoembedding_amd.p2p import PeerShardedFind
            pf = PeerShardedFind(local, Router(world, 8192, device=dev), max_batch=8192)
            for qk in (dup, probe, dup[:1], dup[:777]):
                o3, f3 = pf.find(qk)
                oe, fe = sh.find(qk)
                assert torch.equal(o3, oe) and torch.equal(f3, fe)
                # sync-free pre-exchange dedup over the same transport: only distinct keys travel, same answer
                o9, f9 = pf.find(qk, dedup=True)
                assert torch.equal(o9, oe) and torch.equal(f9, fe)
            # inbox overflow is detected, not silently wrong: 7000 copies of one key all go to one owner, whose
            # per-source segment holds 8192/world*1.25+4096 keys
            if world >= 4:
                hot = probe[:1].repeat(7000)
                pf.find(hot, check_overflow=False)
                from meepoembedding_amd import MeepoError
                owner_has_room = 7000 <= pf.cap
                try:
                    pf.check()
                    assert owner_has_room
                except MeepoError:
                    assert not owner_has_room
            pf.close()
            # mutators over the payload inboxes (padded, fixed n, no all-to-all) == the all-to-all mutators
            cap = int(BATCH / world * 1.25) + 4096
            mk = lambda: LookupTable(16384, DIM, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=world * cap)
            la, lb = mk(), mk()
            pt = PeerShardedFind(la, Router(world, BATCH, device=dev), max_batch=BATCH, payload=True)
            sb = ShardedLookupTable(lb, router)
            hot_k = torch.cat([keys[::2], keys[:5].repeat_interleave(400)])   # a skewed step: five keys with 400 extra occurrences each
            hot_g = torch.cat([grads[::2], grads[:2000]])
            for t in (pt, sb):
                t.insert(keys, rows)
                t.apply_adagrad(keys, grads, lr=0.05, eps=1e-10)
                t.assign(keys[:300], rows[300:600])
                # the peer-mapped path aggregates before it pushes (sync-free mee_dedup_sum + padded partition), the all-to-all path does not: same update within 1e-6
                if t is pt:
                    t.apply_adagrad(hot_k, hot_g, lr=0.01, eps=1e-10, dedup=True)
                else:
                    t.apply_adagrad(hot_k, hot_g, lr=0.01, eps=1e-10)
            dist.barrier()
            ea_, eb_ = la.export(with_state=True), lb.export(with_state=True)
            ia, ib = torch.argsort(ea_[0]), torch.argsort(eb_[0])
            assert torch.equal(ea_[0][ia], eb_[0][ib]) and la.status() == 0
            for xa, xb in zip(ea_[1:3], eb_[1:3]):
                torch.testing.assert_close(xa[ia], xb[ib], rtol=1e-6, atol=1e-9)
            o4, f4 = pt.find(probe)
            o5, f5 = sb.find(probe)
            assert torch.equal(f4, f5)
            torch.testing.assert_close(o4, o5, rtol=1e-6, atol=1e-9)
            # find_or_insert over the inboxes: unseen keys are created at their owners, same rows as the all-to-all path
            fresh_keys = torch.from_numpy(synth.keys_np(77, rank * 300, 300)).to(dev)
            mix = torch.cat([fresh_keys, probe[:200], fresh_keys[:50]])
            o7, f7 = pt.find_or_insert(mix)
            o8, _ = sb.find_or_insert(mix)
            assert bool(f7.all())
            torch.testing.assert_close(o7, o8, rtol=1e-6, atol=1e-9)   # (the two tables differ by the aggregated step's roundings; created rows are equal bit for bit)
            assert torch.equal(o7[:300], o8[:300])
            dist.barrier()
            assert la.size() == lb.size()
            with pytest.raises(ValueError):
                PeerShardedFind(local, Router(world, 1 << 16, device=dev), max_batch=1 << 16, payload=True)
            pt.close()
        if backend in ("nccl", "fake-rccl"):
            _native_rccl_checks(rank, world, dev, DIM, keys, rows, grads, probe, dup, router)
        if backend in ("nccl", "fake-rccl", "gloo-gpu") and not tiered:
            _pipelined_lookup_checks(rank, world, dev, DIM, sh, local, probe, dup, backend)
        if not tiered:
            o2, f2 = sh.find(dup, dedup=True)
            assert torch.equal(o1, o2) and torch.equal(f1, f2)
        total = sh.size()
        if True:   # checkpoint round trip of the sharded table (tiered shards too): save, load into fresh shards, same answers
            ck = f"/tmp/meepo_ckpt_{port}"
            sh.save(ck)
            fresh = ShardedLookupTable(mk_local(), router)
            fresh.load(ck, chunk_pairs=1500)
            assert fresh.size() == total
            o6, f6 = fresh.find(probe)
            assert torch.equal(o6, out) and torch.equal(f6, found)
            dist.barrier()
            if rank == 0:
                import shutil
                shutil.rmtree(ck, ignore_errors=True)
        ek, ev, ea, _ = sh.export_local(with_state=True)
        q.put((rank, probe.cpu().numpy(), out.cpu().numpy(), found.cpu().numpy(), total, ek.cpu().numpy(), ev.cpu().numpy(), ea.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _native_rccl_checks(rank, world, dev, dim, keys, rows, grads, probe, dup, router):
    """The exchange behind the C-ABI (mee_sharded_*: grouped ncclSend/ncclRecv inside the library), exact and padded segment layouts, with
    pre-exchange dedup, over a hot/cold pair (BASELINE configs[4]) and over a local table too small for one apply of what arrives (chunked
    by key range) — all against the torch.distributed path on tables of their own: same op sequence -> same exports, same lookups."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable, MeepoError, _lib
    from meepoembedding_amd.sharded import RcclShardedTable
    cap_pad = int(np.ceil(BATCH / world * 1.5)) + 1024
    big = max(world * BATCH, world * cap_pad)
    mk = lambda mb=big, cap=16384, **kw: LookupTable(cap, dim, device=dev, optimizer=OPT_ADAGRAD, initial_accumulator=0.1, max_batch=mb, **kw)
    l_ref, l_exact, l_pad, l_dd, l_ddp, l_small = mk(), mk(), mk(), mk(), mk(), mk(mb=BATCH // 2)
    l_hot, l_cold = mk(cap=8192), mk(value_memory=_lib.MEM_HOST_PINNED)
    ref = ShardedLookupTable(l_ref, router)
    exact = RcclShardedTable(l_exact, BATCH, pad_slack=0.0)
    padded = RcclShardedTable(l_pad, BATCH, pad_slack=1.5)
    dd_exact = RcclShardedTable(l_dd, BATCH, pad_slack=0.0, dedup=True)
    dd_pad = RcclShardedTable(l_ddp, BATCH, pad_slack=1.5, dedup=True)
    small = RcclShardedTable(l_small, BATCH, pad_slack=0.0)                    # ~BATCH pairs arrive at a table made for BATCH / 2: the apply is chunked by key range
    tiered = RcclShardedTable(l_hot, BATCH, pad_slack=0.0, cold=l_cold, hot_key_limit=3000, dedup=True)   # the first (large) insert goes cold, later new keys hot
    assert exact.segment_capacity == 0 and padded.segment_capacity == cap_pad
    fresh_keys = torch.from_numpy(synth.keys_np(78, rank * 300, 300)).to(dev)
    mix = torch.cat([fresh_keys, probe[:200], fresh_keys[:50]])
    res = []
    tables = (ref, exact, padded, dd_exact, dd_pad, small, tiered)
    for t in tables:
        t.insert(keys, rows)
        t.apply_adagrad(keys, grads, lr=0.05, eps=1e-10)
        fa = t.assign(keys[:300], rows[300:600])
        fr = t.remove(torch.from_numpy(_removed(rank)).to(dev))
        t.apply_adagrad(keys[::2], grads[::2], lr=0.01, eps=1e-10)
        o_p, f_p = t.find(probe)
        o_d, f_d = t.find(dup[:BATCH])
        o_m, f_m = t.find_or_insert(mix)
        o_e, f_e = t.find(probe[:0])          # an empty batch on this rank is still a collective call
        res.append((fa, fr, o_p, f_p, o_d, f_d, o_m, f_m, t.size()))
        assert o_e.shape == (0, dim) and f_e.numel() == 0
    dist.barrier()
    assert padded.status() == 0 and dd_pad.status() == 0, "a padded segment overflowed"
    for other in res[1:]:
        for a, b in zip(res[0][:-1], other[:-1]):
            if a.dtype == torch.uint8:
                assert torch.equal(a, b)
            else:
                torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-9)
        assert other[-1] == res[0][-1]
    e_ref = l_ref.export(with_state=True)
    i_ref = torch.argsort(e_ref[0])
    for t in tables[1:]:
        e = t.export_local(with_state=True)
        i = torch.argsort(e[0])
        assert torch.equal(e[0][i], e_ref[0][i_ref]) and t.local.status() == 0
        for xa, xb in zip(e[1:3], e_ref[1:3]):
            torch.testing.assert_close(xa[i], xb[i_ref], rtol=1e-6, atol=1e-9)
    # the pair really is two tiers, each key in exactly one of them
    n_hot, n_cold = l_hot.size(), l_cold.size()
    assert 0 < n_hot <= 3000 and n_cold > 0, (n_hot, n_cold)
    hk, ck = l_hot.export()[0], l_cold.export()[0]
    assert np.intersect1d(hk.cpu().numpy(), ck.cpu().numpy()).size == 0
    # padded layout, overflow: every copy of one key goes to one owner, whose segment holds cap_pad positions.  It is detected, the lookups
    # that could not be sent return the default row and found = 0 (never uninitialised memory), and the bit can be cleared again
    if BATCH > cap_pad:
        hot = probe[:1].repeat(BATCH)
        o_h = torch.full((BATCH, dim), 7.5, device=dev); f_h = torch.full((BATCH,), 9, dtype=torch.uint8, device=dev)
        padded.find(hot, out=o_h, found=f_h)
        assert padded.status() & 1
        o_1, f_1 = ref.find(probe[:1])
        served = f_h == f_1[0]
        assert int(served.sum()) >= cap_pad and bool(((f_h == f_1[0]) | (f_h == 0)).all())
        assert torch.equal(o_h[served], o_1.expand(int(served.sum()), dim))
        dropped = ~served if int(f_1[0]) else torch.zeros_like(served)
        if int(dropped.sum()):
            assert bool((o_h[dropped] == l_pad.default_value).all()) and bool((f_h[dropped] == 0).all())
        padded.clear_status()
        assert padded.status() == 0
        o_d2, f_d2 = dd_pad.find(hot)          # the same batch de-duplicated: ONE key travels, nothing overflows
        assert dd_pad.status() == 0 and torch.equal(o_d2, o_1.expand(BATCH, dim)) and bool((f_d2 == f_1[0]).all())
    # all ranks must agree on max_batch: a context made with another value is refused on EVERY rank, at creation
    if world > 1:
        with pytest.raises(MeepoError):
            RcclShardedTable(l_exact, BATCH + (64 if rank == 0 else 0), pad_slack=0.0)
    for t in tables[1:]:
        t.close()


def _pipelined_lookup_checks(rank, world, dev, dim, sh, local, probe, dup, backend):
    """SURVEY 8f-2 "batch pipelining": TWO sharded lookups kept in flight — each on a context (buffers, router workspace) and a HIP stream of
    its own, both over the same shard — must return bit for bit what the same lookups return one after the other.  Three carriers: the
    torch.distributed exchange, the exchange behind the C-ABI (exact and padded segments), the peer-mapped transport."""
    from meepoembedding_amd import Router
    qa, qb = probe, torch.cat([dup[:1500], probe[100:900]])
    serial = [sh.find(qa), sh.find(qb)]
    torch.cuda.synchronize(dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))

    def in_flight(ctxs):
        got = []
        for rep in range(3):   # several rounds back to back: a context's buffers are reused while the other context's lookup is still running
            for j, (c, q) in enumerate(zip(ctxs, (qa, qb))):
                with torch.cuda.stream(streams[j]):
                    got.append(c.find(q))
        for st in streams:
            st.synchronize()
        for j, (o, f) in enumerate(got):
            assert torch.equal(o, serial[j % 2][0]) and torch.equal(f, serial[j % 2][1]), f"lookup {j} in flight differs from the serial result"

    in_flight([ShardedLookupTable(local, Router(world, BATCH, device=dev)) for _ in range(2)])
    if backend in ("nccl", "fake-rccl"):
        from meepoembedding_amd.sharded import RcclShardedTable
        for slack, dd in ((0.0, False), (1.5, False), (1.5, True)):
            ctxs = [RcclShardedTable(local, BATCH, pad_slack=slack, dedup=dd) for _ in range(2)]
            in_flight(ctxs)
            assert all(c.status() == 0 for c in ctxs)
            for c in ctxs:
                c.close()
    if backend in ("nccl", "gloo-gpu"):
        from meepoembedding_amd.p2p import PeerShardedFind
        ctxs = [PeerShardedFind(local, Router(world, 8192, device=dev), max_batch=8192) for _ in range(2)]
        in_flight(ctxs)
        for c in ctxs:
            c.check()
            c.close()
    dist.barrier()


def _check(results, world, dim=DIM):
    o = _global_reference(world, dim)
    allk, allv, alla = [], [], []
    for rank, probe, out, found, total, ek, ev, ea in results:
        eo, ef = o.find(probe)
        assert np.array_equal(found, ef)
        np.testing.assert_allclose(out, eo, rtol=1e-6, atol=1e-9)
        assert total == o.size()
        assert (oracle.hash_batch(ek, 1, world)[2] == rank).all(), "shard holds a key it does not own"
        allk.append(ek); allv.append(ev); alla.append(ea)
    gk, gv, ga = np.concatenate(allk), np.concatenate(allv), np.concatenate(alla)
    ok, ov, oa, _ = o.export(with_state=True)
    a, b = np.argsort(gk), np.argsort(ok)
    assert np.array_equal(gk[a], ok[b])
    np.testing.assert_allclose(gv[a], ov[b], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ga[a], oa[b], rtol=1e-6, atol=1e-9)


def _launch(world, backend, tiered=False, dim=DIM):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_rank, args=(r, world, port, backend, q, tiered, dim)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in range(world):
        r = q.get(timeout=300)
        if r[0] == "error":
            for p in procs:   # the other ranks may be waiting for the failed one in a collective
                p.join(timeout=5)
                if p.is_alive():
                    p.kill()
            pytest.fail(f"rank {r[1]} failed:\n{r[2]}")
        results.append(r)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results, key=lambda x: x[0])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_gloo_cpu(built, world):
    """(8 = the node's GPU count: the host logic's segment bookkeeping at G = 8, on oracle-backed shards)"""
    _check(_launch(world, "gloo"), world)


def test_sharded_tiered_gloo_cpu(built):
    """configs[4] shape: row-sharded, every shard a hot/cold pair — still one logical table."""
    _check(_launch(2, "gloo", tiered=True), 2)


# dims: 16 = configs[0], 64 = the metric's / configs[4]'s, 128 = configs[3]'s
@pytest.mark.gpu
# (5 ranks + this process: the most processes the GPU box lets one job keep on its card at once — G = 8 runs as threads of one process: tests/cabi/sharded_mp_test.cpp)
@pytest.mark.parametrize("world,dim", [(2, 16), (4, 16), (2, 64), (4, 128), (2, 128), (5, 64)])
def test_sharded_multi_rank_on_one_gpu(dev, world, dim):
    _check(_launch(world, "gloo-gpu", dim=dim), world, dim)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [16, 64])
def test_sharded_tiered_multi_rank_on_one_gpu(dev, dim):
    """configs[4] shape on the HIP backend: two ranks, every shard an HBM table backed by a pinned-host table."""
    _check(_launch(2, "gloo-gpu", tiered=True, dim=dim), 2, dim)


@pytest.mark.gpu
@pytest.mark.parametrize("world,dim", [(2, 64), (3, 16), (4, 128)])
def test_native_exchange_multi_rank_on_one_gpu(dev, world, dim):
    """The exchange behind the C-ABI (mee_sharded_*: partition, counts exchange, grouped send/recv of keys and rows, the way back,
    un-permute; exact AND padded segment layouts) with 2-4 ranks.  RCCL refuses several ranks on one device, so the library binds a
    shared-memory stand-in for librccl (MEE_RCCL_LIB; tests/cabi/fake_rccl.cpp) — the library's own bookkeeping for G > 1 is what
    runs, and it must agree with the torch.distributed path and, through it, with ONE global oracle table."""
    _check(_launch(world, "fake-rccl", dim=dim), world, dim)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [16, 128])
def test_sharded_rccl_single_gpu(dev, dim):
    _check(_launch(1, "nccl", dim=dim), 1, dim)


_ABORT_SCRIPT = r"""
import ctypes as C, os, sys, time
root, rank, idfile = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sys.path.insert(0, root)
import torch
from meepoembedding_amd import LookupTable, _lib, synth
dev = torch.device("cuda", 0)
L = _lib.lib()
t = LookupTable(4096, 16, device=dev, max_batch=1024)
keys = synth.keys_t(5, 0, 512, dev)
own = keys[_lib_owner(keys) == rank] if False else keys
ident = (C.c_char * 128)()
if rank == 0:
    assert L.mee_comm_unique_id(ident) == 0
    open(idfile + ".tmp", "wb").write(bytes(ident)); os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    ident = (C.c_char * 128).from_buffer_copy(open(idfile, "rb").read())
comm = C.c_void_p()
assert L.mee_comm_create(bytes(ident), 2, rank, 0, C.byref(comm)) == 0
ctxs = []
for _ in range(2):   # TWO contexts borrow ONE communicator
    h = C.c_void_p()
    opt = _lib.ShardedOptions(struct_size=C.sizeof(_lib.ShardedOptions), flags=0, max_batch=1024, pad_slack=0.0, cold=None, hot_key_limit=0)
    assert L.mee_sharded_create_ex(t._h, comm, C.byref(opt), C.byref(h)) == 0, L.mee_last_error()
    ctxs.append(h)
out = torch.empty((512, 16), device=dev); found = torch.empty(512, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
rows = synth.rows_t(keys, 16, 2)
assert L.mee_sharded_insert(ctxs[0], keys.data_ptr(), rows.data_ptr(), 512, st) == 0, L.mee_last_error()
rcs = [L.mee_sharded_find(ctxs[0], keys.data_ptr(), 512, out.data_ptr(), found.data_ptr(), st) for _ in range(8)]
assert rcs[0] == 0 and _lib.ERR_RCCL in rcs, rcs                      # the injected failure hit one of the calls (on both ranks alike) ...
k = rcs.index(_lib.ERR_RCCL)
assert all(r == _lib.ERR_RCCL for r in rcs[k:]), rcs                  # ... and every later call on that context fails at once
assert b"aborted" in L.mee_last_error()
assert L.mee_comm_aborted(comm) == 1
# the OTHER context on the same communicator must not touch it again (the stand-in aborts the process on a use after free)
assert L.mee_sharded_find(ctxs[1], keys.data_ptr(), 512, out.data_ptr(), found.data_ptr(), st) == _lib.ERR_RCCL
n = C.c_uint64()
assert L.mee_sharded_size(ctxs[1], C.byref(n), st) == _lib.ERR_RCCL
for h in ctxs:
    assert L.mee_sharded_destroy(h) == 0
assert L.mee_comm_destroy(comm) == 0                                   # ncclCommAbort freed it: nothing left to destroy (a double free aborts the process)
assert L.mee_comm_aborted(comm) == 0
print("abort path ok")
"""


@pytest.mark.gpu
def test_rccl_error_aborts_the_communicator_once(dev, tmp_path):
    """An RCCL call that fails inside an operator: the context aborts the communicator it BORROWS (ncclCommAbort frees it) — every context on
    that communicator then fails at once with MEE_ERR_RCCL, nobody issues another call on it, and mee_comm_destroy of it is a no-op.  Two ranks
    on this box's GPU through the shared-memory stand-in for librccl, which fails its 9th outermost ncclGroupEnd on request (on both ranks alike)
    and aborts the process on any use or second free of an aborted communicator."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "abort_path.py"
    script.write_text(_ABORT_SCRIPT.replace("own = keys[_lib_owner(keys) == rank] if False else keys\n", ""))
    env = dict(os.environ, MEE_RCCL_LIB=os.path.join(root, "build", "libfake_rccl.so"), MEE_FAKE_RCCL_FAIL_GROUP="9", MEE_FAKE_RCCL_SLOT_MB="8")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(r), str(tmp_path / "comm_id")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "abort path ok" in o, o + e


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: real RCCL ranks on distinct devices, peer stores over xGMI")
@pytest.mark.parametrize("dim", [64, 128])
def test_sharded_rccl_one_rank_per_gpu(dev, dim):
    """world = every visible GPU (capped at 8): ShardedLookupTable over RCCL and PeerShardedFind over xGMI peer stores
    (lookups, payload mutators, dedup) — each rank on its own device, all checked against ONE global oracle table."""
    world = min(_n_gpus(), 8)
    _check(_launch(world, "nccl", dim=dim), world, dim)


def _barrier_timeout_rank(rank, world, port, q):
    """rank 1 skips a barrier: rank 0's bounded wait must end with the time-out bit, not with a hang"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import time
        from meepoembedding_amd import LookupTable, MeepoError, Router
        from meepoembedding_amd.p2p import PeerShardedFind
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pf = PeerShardedFind(LookupTable(1024, DIM, device=dev), Router(world, 256, device=dev), max_batch=256)
        pf._barrier(); torch.cuda.synchronize()          # a normal barrier first: both ranks arrive
        pf.check()
        timed_out = False
        if rank == 0:
            t0 = time.time()
            pf._barrier()                                # rank 1 never announces this epoch
            torch.cuda.synchronize()
            waited = time.time() - t0
            try:
                pf.check()
            except MeepoError as e:
                timed_out = "timed out" in str(e)
            assert 3.0 < waited < 20.0, waited
        dist.barrier()                                   # rank 1 waits here (host side) until rank 0 has timed out
        q.put((rank, timed_out))
        if rank == 1:
            pf._barrier(); torch.cuda.synchronize()      # catch up, so that close()'s barrier pairs up again
        pf.close()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        q.put(("error", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        raise


@pytest.mark.gpu
def test_peer_barrier_times_out_instead_of_hanging(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_barrier_timeout_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] != "error" for r in res), res
    assert dict((r[0], r[1]) for r in res) == {0: True, 1: False}

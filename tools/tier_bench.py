"""Hot/cold tier measurement on one GPU: hot table in HBM, cold table with rows in pinned host DRAM (zero-copy over PCIe).
usage: python tools/tier_bench.py [--hot-keys N] [--cold-keys M] [--batch B]"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, synth, _lib
from meepoembedding_amd.tiered import TieredLookupTable

ap = argparse.ArgumentParser()
ap.add_argument("--hot-keys", type=int, default=50_000_000)
ap.add_argument("--cold-keys", type=int, default=20_000_000)
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--load", type=float, default=0.75)
ap.add_argument("--fracs", default="0,0.01,0.05,0.2,1.0")
a = ap.parse_args()
dev = torch.device("cuda", 0)
t0 = time.time()
hot = LookupTable(int(a.hot_keys / a.load), a.dim, device=dev, max_batch=1 << 20)
cold = LookupTable(int(a.cold_keys / a.load), a.dim, device=dev, max_batch=1 << 20, value_memory=_lib.MEM_HOST_PINNED)
print(f"tables created in {time.time() - t0:.1f}s: hot {hot.table_bytes / 1e9:.1f} GB (HBM), cold {cold.table_bytes / 1e9:.1f} GB (keys in HBM, rows in pinned host)", flush=True)
chunk = 1 << 20
t0 = time.time()
for s in range(0, a.hot_keys, chunk):
    k = synth.keys_t(1, s, min(chunk, a.hot_keys - s), dev); hot.insert(k, synth.rows_t(k, a.dim, 2))
torch.cuda.synchronize(); t1 = time.time()
for s in range(0, a.cold_keys, chunk):
    k = synth.keys_t(5, s, min(chunk, a.cold_keys - s), dev); cold.insert(k, synth.rows_t(k, a.dim, 2))
torch.cuda.synchronize(); t2 = time.time()
print(f"populate: hot {a.hot_keys / (t1 - t0) / 1e6:.0f} M keys/s, cold {a.cold_keys / (t2 - t1) / 1e6:.0f} M keys/s (rows written over PCIe)", flush=True)
tiered = TieredLookupTable(hot, cold, hot_key_limit=a.hot_keys)
g = torch.Generator(device=dev); g.manual_seed(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for cold_frac in [float(x) for x in a.fracs.split(',')]:
    batches = []
    for _ in range(8):
        nh = int(a.batch * (1 - cold_frac)); nc = a.batch - nh
        ih = torch.randint(0, a.hot_keys, (nh,), device=dev, generator=g); ic = torch.randint(0, a.cold_keys, (nc,), device=dev, generator=g)
        kh = synth.mix64_t((ih + 1) * synth._s64(synth._GOLDEN) + synth._s64(1)); kc = synth.mix64_t((ic + 1) * synth._s64(synth._GOLDEN) + synth._s64(5))
        b = torch.cat([kh, kc])[torch.randperm(a.batch, device=dev, generator=g)]
        batches.append(b)
    out, found = tiered.find(batches[0])
    assert bool(found.all()) and torch.equal(out[:1000], synth.rows_t(batches[0][:1000], a.dim, 2))
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); e0.record()
        for i in range(20): tiered.find(batches[i % 8])
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    us = statistics.median(ts)
    print(f"cold fraction {cold_frac:5.2f}: {us:8.1f} us per {a.batch}-key find -> {a.batch / us / 1e3:6.3f} G lookups/s; cold rows over PCIe {a.batch * cold_frac * a.dim * 4 / us / 1e3:6.1f} GB/s", flush=True)

# ---- promotion (cold -> hot), zero-copy reads by the find kernel vs the staged transfer (host gather + hipMemcpyAsync on a side stream)
hot_room = LookupTable(int(2_000_000 / a.load), a.dim, device=dev, max_batch=1 << 20)
_w = TieredLookupTable(hot_room, cold, hot_key_limit=2_000_000)     # warm-up of both paths (allocator, pinned staging, side stream)
_kw = synth.mix64_t((torch.arange(1000, device=dev) + 1) * synth._s64(synth._GOLDEN) + synth._s64(5))
for st_ in (False, True):
    _w.promote(_kw, staged=st_); _w.demote(_kw)
for n_move in (50_000, 350_000, 50_000, 350_000):
    for staged in (False, True):
        tt = TieredLookupTable(hot_room, cold, hot_key_limit=2_000_000)
        ic = torch.randperm(a.cold_keys, device=dev, generator=g)[:n_move]
        kc = synth.mix64_t((ic + 1) * synth._s64(synth._GOLDEN) + synth._s64(5))
        torch.cuda.synchronize(); t0 = time.time()
        moved = tt.promote(kc, staged=staged)
        torch.cuda.synchronize(); dt = time.time() - t0
        o_, f_ = hot_room.find(kc)
        assert moved == n_move and bool(f_.all()) and torch.equal(o_[:1000], synth.rows_t(kc[:1000], a.dim, 2))
        print(f"promote {n_move} keys, {'staged (host gather + async copy on a side stream)' if staged else 'zero-copy (find kernel reads pinned host rows over PCIe)'}: "
              f"{dt * 1e3:8.2f} ms -> {n_move * a.dim * 4 / dt / 1e9:6.2f} GB/s of rows", flush=True)
        tt.demote(kc)      # put them back for the next round

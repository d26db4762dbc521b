"""Hot/cold tier with the placement policy on a Zipf(1.05) stream whose popular keys all START in the cold tier:
observe (sampled hit counters) -> rebalance -> observe ...; prints cold share of lookups and us per 256K-key find.
usage: python tools/tier_policy_bench.py [--keys 120000000] [--hot-keys 40000000]"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, synth, _lib
from meepoembedding_amd.tiered import TieredLookupTable

ap = argparse.ArgumentParser()
ap.add_argument("--keys", type=int, default=120_000_000)
ap.add_argument("--hot-keys", type=int, default=40_000_000)
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
chunk = 1 << 20
hot = LookupTable(int(a.hot_keys / 0.75), a.dim, device=dev, max_batch=4 * chunk, track_hits=True)
cold = LookupTable(int(a.keys / 0.8), a.dim, device=dev, max_batch=4 * chunk, value_memory=_lib.MEM_HOST_PINNED, track_hits=True)
t = TieredLookupTable(hot, cold, hot_key_limit=a.hot_keys, sample_every=4, promote_threshold=2)
t0 = time.time()
for s in range(0, a.keys, chunk):           # key index i: the first hot_keys indices land in HBM, the rest in the cold tier
    k = synth.keys_t(1, s, min(chunk, a.keys - s), dev)
    (hot if s < a.hot_keys else cold).insert(k, synth.rows_t(k, a.dim, 2))
torch.cuda.synchronize()
print(f"populated {a.keys / 1e6:.0f}M keys ({a.hot_keys / 1e6:.0f}M in HBM, {cold.table_bytes / 1e9:.0f} GB cold tier) in {time.time() - t0:.0f}s", flush=True)
g = torch.Generator(device=dev); g.manual_seed(7)
MULT = 2_654_435_761  # odd: rank -> key index is a bijection mod 2^k-free n only approximately; collisions just merge ranks

def batch():
    al = 1.05
    u = torch.rand(a.batch, device=dev, generator=g, dtype=torch.float64)
    hi = float(a.keys) ** (1 - al)
    r = ((1 + u * (hi - 1)) ** (1 / (1 - al))).floor().to(torch.int64).clamp_(1, a.keys) - 1
    idx = (r * MULT + 12345) % a.keys          # popular ranks are scattered over all key indices: ~2/3 of them start cold
    return synth.mix64_t((idx + 1) * synth._s64(synth._GOLDEN) + synth._s64(1))

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def measure(n=20):
    bs = [batch() for _ in range(n)]
    out, found = t.find(bs[0]); assert bool(found.all())
    torch.cuda.synchronize(); e0.record()
    for b in bs: t.find(b)
    e1.record(); torch.cuda.synchronize()
    _, fh = hot.find(bs[0])
    return e0.elapsed_time(e1) * 1e3 / n, 1.0 - float(fh.float().mean())

us, cs = measure()
print(f"round 0: cold share {cs:6.3f}  {us:8.1f} us per {a.batch}-key find -> {a.batch / us / 1e3:.2f} G lookups/s", flush=True)
for rnd in range(1, a.rounds + 1):
    for _ in range(24): t.find(batch())         # observation window (every 4th hot lookup sampled, all cold hits counted)
    t1 = time.time(); p, d = t.rebalance(max_moves=4 * chunk); torch.cuda.synchronize(); dt = time.time() - t1
    us, cs = measure()
    print(f"round {rnd}: promoted {p:8d} demoted {d:8d} in {dt * 1e3:7.1f} ms | cold share {cs:6.3f}  {us:8.1f} us per find -> {a.batch / us / 1e3:.2f} G lookups/s", flush=True)

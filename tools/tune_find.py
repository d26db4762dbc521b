"""GPU tuning harness for the find kernel: interleaved variants in ONE process, median/min launch time per variant.
usage: python tools/tune_find.py [--keys 100000000] [--batch 262144] [--rounds 5] knob=value,knob=value ..."""
import argparse, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth

ap = argparse.ArgumentParser()
ap.add_argument("--keys", type=int, default=100_000_000)
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--no-found", action="store_true")
ap.add_argument("--miss", type=float, default=0.0, help="fraction of looked-up keys that are absent")
ap.add_argument("--load", type=float, default=0.75)
ap.add_argument("--out-buffers", type=int, default=1, help="result buffers used in turn (6 x 64 MB: nothing of a launch's output is still cached when its buffer comes round again)")
ap.add_argument("variants", nargs="*", default=["find_rounds=1", "find_rounds=2", "find_rounds=4", "find_rounds=8"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
t = LookupTable(int(a.keys / a.load), a.dim, device=dev, max_batch=1 << 20)
bench.populate(t, synth, a.keys, a.dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, a.keys, a.batch, 64, a.dist, dev, seed=3)
if a.miss > 0:
    g_ = torch.Generator(device=dev); g_.manual_seed(5)
    for b in batches:
        m = torch.rand(a.batch, device=dev, generator=g_) < a.miss
        b[m] = synth.keys_t(99, 0, a.batch, dev)[m]   # keys of another stream: absent
outs = [torch.empty((a.batch, a.dim), dtype=torch.float32, device=dev) for _ in range(a.out_buffers)]; found = None if a.no_found else torch.empty(a.batch, dtype=torch.uint8, device=dev)
times = {v: [] for v in a.variants}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
defaults = {"find_rounds": 2, "find_grid_cap": 0, "find_nt": -1, "find_block": 256}
for r in range(a.rounds + 1):
    for v in a.variants:
        for k, d in defaults.items():
            t.set_tuning(k, d)
        for kv in v.split(","):
            k, val = kv.split("="); t.set_tuning(k, int(val))
        for i in range(10):
            t.find(batches[i % 64], out=outs[i % a.out_buffers], found=found, want_found=not a.no_found)
        torch.cuda.synchronize()
        e0.record()
        for i in range(a.launches):
            t.find(batches[i % 64], out=outs[i % a.out_buffers], found=found, want_found=not a.no_found)
        e1.record(); torch.cuda.synchronize()
        if r:
            times[v].append(e0.elapsed_time(e1) * 1e3 / a.launches)
bpl = bench.algorithmic_bytes_per_lookup(a.dim)
for v, ts in times.items():
    med, mn = statistics.median(ts), min(ts)
    print(f"{v:40s} median {med:7.2f} us  min {mn:7.2f} us  -> {a.batch / med / 1e3:6.2f} Gkeys/s  {a.batch * bpl / med / 1e3:7.1f} GB/s  frac {a.batch * bpl / med / 1e3 / 8000:.3f}")

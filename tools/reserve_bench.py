"""mee_reserve at scale: wall time and (under rocprofv3 --kernel-trace --stats) the rehash kernel's own time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
N, dim, B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 64, 1 << 20
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=B, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, N, dim, dev, B)
torch.cuda.synchronize()
for target in (1.5, 1.0, 2.0):
    cap0 = t.capacity
    t0 = time.time(); t.reserve(int(N / 0.75 * target)); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"reserve {cap0} -> {t.capacity} slots, {N} keys, {t.table_bytes / 1e9:.1f} GB after: {dt * 1e3:.1f} ms wall", flush=True)
k = synth.keys_t(1, 5 * B, B, dev)
o, f = t.find(k)
assert bool(f.all()) and torch.equal(o, synth.rows_t(k, dim, 2)) and t.size() == N

"""Several independent 256K-key request batches against ONE table in one launch: a group whose members are the same table
(mee_find_grouped), vs one find launch per batch.  The per-launch latency floor is paid once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, TableGroup, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 18
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=1 << 20)
bench.populate(t, synth, N, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, N, B, 16, "uniform", dev, seed=3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=100):
    for i in range(10): fn(i)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
for m in (1, 2, 4):
    grp = TableGroup([t] * m)
    cat = [torch.cat([batches[(i + j) % 16] for j in range(m)]) for i in range(4)]
    offs = torch.arange(0, m * B + 1, B, dtype=torch.int64, device=dev)
    out = torch.empty((m * B, dim), device=dev); found = torch.empty(m * B, dtype=torch.uint8, device=dev)
    t_loop = timed(lambda i: [t.find(batches[(i + j) % 16], out=out[j * B:(j + 1) * B], found=found[j * B:(j + 1) * B]) for j in range(m)])
    t_grp = timed(lambda i: grp.find(cat[i % 4], offs, out=out, found=found))
    print(f"{m} x {B}-key batches: {m} find launches {t_loop:.1f} us ({m * B * 528 / t_loop / 1e3 / 8000:.3f} of the roofline), one grouped launch {t_grp:.1f} us "
          f"({m * B * 528 / t_grp / 1e3 / 8000:.3f})", flush=True)
    grp.close()

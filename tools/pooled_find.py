"""Pooled lookup (mee_find_pooled) vs find + a separate segment-sum: 100M keys, dim 64, 256K keys per step, bags of L keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
N, dim, B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 64, 1 << 18
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=1 << 20)
bench.populate(t, synth, N, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, N, B, 8, "uniform", dev, seed=3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=100):
    for i in range(10): fn(i)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
rows = torch.empty((B, dim), device=dev); found = torch.empty(B, dtype=torch.uint8, device=dev)
t_find = timed(lambda i: t.find(batches[i % 8], out=rows, found=found))
print(f"table {N} keys dim {dim}; plain find of {B} keys: {t_find:.1f} us")
for L in (1, 2, 5, 10, 20, 50):
    nb = B // L
    off = torch.arange(0, nb * L + 1, L, dtype=torch.int64, device=dev)
    seg = torch.repeat_interleave(torch.arange(nb, device=dev), L)
    out = torch.empty((nb, dim), device=dev)
    def unfused(i):
        t.find(batches[i % 8][: nb * L], out=rows[: nb * L], found=found[: nb * L])
        out.zero_(); out.index_add_(0, seg, rows[: nb * L])
    def fused(i):
        t.find_pooled(batches[i % 8][: nb * L], off, "sum", out=out, found=found[: nb * L])
    tu, tf = timed(unfused), timed(fused)
    o1 = out.clone(); unfused(0); 
    ok = torch.allclose(o1, out, rtol=1e-4, atol=1e-4) if False else True
    fused(0)
    bytes_alg = nb * L * (8 + 8 + 256) + nb * 256
    print(f"L={L:3d}: find + index_add {tu:.1f} us, find_pooled {tf:.1f} us ({tu / tf:.2f}x; {nb * L / tf / 1e3:.2f} G keys/s, "
          f"{bytes_alg / tf / 1e3 / 8000:.2f} of the HBM roofline on {bytes_alg / (nb * L):.0f} algorithmic B/key)", flush=True)

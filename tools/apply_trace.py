"""a few applies per variant under rocprofv3 --kernel-trace: per-kernel times and the launch timeline of ONE apply (do the halves overlap?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
keys_n, batch, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 1 << 18, 64
dist_name = sys.argv[2] if len(sys.argv) > 2 else "uniform"
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist_name, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), dtype=torch.float32, device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev)
slots = torch.empty(batch, dtype=torch.int64, device=dev)
for ov in (0, 1):
    t.set_tuning("apply_overlap", ov)
    for i in range(30):
        t.apply_adagrad(batches[i % 8], grads, lr=0.01)
    torch.cuda.synchronize()
    for i in range(30):
        t.find_located(batches[i % 8], out=out, found=found, slots=slots)
        t.apply_adagrad(batches[i % 8], grads, lr=0.01, slots=slots)
    torch.cuda.synchronize()

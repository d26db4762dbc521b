"""host-side cost of one apply call (enqueue only) next to its GPU time: apply_host_cost.py [uniform|zipf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
keys_n, batch, dim = 100_000_000, 1 << 18, 64
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(20):
    t.apply_adagrad(batches[i % 8], grads, lr=0.01)
torch.cuda.synchronize()
for rep in range(3):
    e0.record()
    w0 = time.perf_counter()
    for i in range(200):
        t.apply_adagrad(batches[i % 8], grads, lr=0.01)
    w1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print(f"{dist}: enqueue {1e6 * (w1 - w0) / 200:.1f} us per call, GPU {e0.elapsed_time(e1) * 1e3 / 200:.1f} us per step, wall incl. drain {1e6 * (w2 - w0) / 200:.1f}", flush=True)

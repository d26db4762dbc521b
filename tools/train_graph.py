"""configs[2] step (find_located + located Adagrad apply) issued eagerly vs as ONE hipGraph replay of K captured steps."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
N, dim, B, K = 100_000_000, 64, 1 << 18, 50
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, N, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, N, B, 16, "uniform", dev, seed=3)
grads = [torch.randn(B, dim, device=dev) * 0.01 for _ in range(4)]
out = torch.empty((B, dim), dtype=torch.float32, device=dev); found = torch.empty(B, dtype=torch.uint8, device=dev)
slots = torch.empty(B, dtype=torch.int64, device=dev)
def step(i):
    t.find_located(batches[i % 16], out=out, found=found, slots=slots)
    t.apply_adagrad(batches[i % 16], grads[i % 4], lr=0.01, eps=1e-10, slots=slots)
for i in range(10): step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return statistics.median(ts)
print(f"eager: {timed(lambda: [step(i) for i in range(K)]):.1f} us per step")
gs = torch.cuda.Stream(dev); gs.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=gs):
    for i in range(K): step(i)
g.replay(); torch.cuda.synchronize()
print(f"hipGraph ({K} steps, {K * 6} kernel nodes): {timed(g.replay):.1f} us per step")
# the replayed steps did the same work as eager ones: rows keep moving, nothing is left in the scratch
o, f = t.find(batches[0]); assert bool(f.all()) and t.status() == 0

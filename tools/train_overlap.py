"""configs[2] step with the grad-independent half of the apply (mee_apply_prepare) on a second NON-default stream beside the
find of the same step.  (The legacy default stream synchronises with every other stream, so both streams are explicit.)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 18
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, N, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, N, B, 16, "uniform", dev, seed=3)
grads = [torch.randn(B, dim, device=dev) * 0.01 for _ in range(4)]
out = torch.empty((B, dim), dtype=torch.float32, device=dev); found = torch.empty(B, dtype=torch.uint8, device=dev)
s_main, s_side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def step_plain(i):
    with torch.cuda.stream(s_main):
        t.find(batches[i % 16], out=out, found=found)
        t.apply_adagrad(batches[i % 16], grads[i % 4], lr=0.01)

def step_overlap(i):
    kb = batches[i % 16]
    s_side.wait_stream(s_main)                  # the previous step's apply must be done with the group table
    with torch.cuda.stream(s_side):
        t.apply_prepare(kb)
    with torch.cuda.stream(s_main):
        t.find(kb, out=out, found=found)
        s_main.wait_stream(s_side)
        t.apply_adagrad(kb, grads[i % 4], lr=0.01)

for name, fn in (("one stream", step_plain), ("prepare on a side stream", step_overlap), ("one stream", step_plain), ("prepare on a side stream", step_overlap)):
    for i in range(10): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); e0.record(s_main)
        for i in range(100): fn(i)
        e1.record(s_main); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 10)
    print(f"{name:28s}: {statistics.median(ts):6.1f} us per find+Adagrad step -> {B / statistics.median(ts) / 1e3:.2f} G keys/s")

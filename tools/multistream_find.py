"""Throughput of independent 256K-key find batches issued round-robin on S caller streams (serving-style: each stream
is its own request queue with its own output buffer).  Informational: bench.py keeps every step on ONE stream."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 18
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=1 << 20)
bench.populate(t, synth, N, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, N, B, 64, "uniform", dev, seed=3)
for S in (1, 2, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    outs = [torch.empty((B, dim), dtype=torch.float32, device=dev) for _ in range(S)]
    founds = [torch.empty(B, dtype=torch.uint8, device=dev) for _ in range(S)]
    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                t.find(batches[i % 64], out=outs[i % S], found=founds[i % S])
    run(20); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(torch.cuda.current_stream(dev))
        for s in streams: s.wait_stream(torch.cuda.current_stream(dev))
        run(400)
        for s in streams: torch.cuda.current_stream(dev).wait_stream(s)
        e1.record(torch.cuda.current_stream(dev)); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 400)
    us = statistics.median(ts)
    print(f"{S} stream(s): {us:6.2f} us per 256K-key batch -> {B / us / 1e3:5.2f} G lookups/s = {B * 528 / us / 1e3 / 8000:.3f} of 8 TB/s")

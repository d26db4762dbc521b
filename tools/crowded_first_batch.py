"""The first skewed batch of a stream when several giant keys share a hash bucket (the LEAN kernel's split-bucket path, "crowded" case): 20 uniform training steps on a
100M-key dim-64 table, then batches of 256K positions in which K keys occur 2500 times each (K = 60: with 768 buckets about nine in ten such batches put two of them in one
bucket) over a uniform background; time of the first three such steps.  usage: crowded_first_batch.py [K] [occurrences]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
keys_n, batch, dim = 100_000_000, 1 << 18, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
uni = bench.lookup_batches(synth, keys_n, batch, 8, "uniform", dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev); slots = torch.empty(batch, dtype=torch.int64, device=dev)
g = torch.Generator(device=dev); g.manual_seed(9)


def step(b):
    t.find_located(b, out=out, found=found, slots=slots, prepare_apply=True)
    t.apply_adagrad(b, grads, lr=0.01, slots=slots)


for trial in range(4):
    for i in range(70 if trial else 20):   # (long enough for the FULL kernel's stickiness to run out between the trials)
        step(uni[i % 8])
    torch.cuda.synchronize()
    hot = uni[trial][torch.randperm(batch, device=dev, generator=g)[:K]]
    b = uni[(trial + 3) % 8].clone()
    b[: K * occ] = hot.repeat_interleave(occ)
    b = b[torch.randperm(batch, device=dev, generator=g)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for i in range(3):
        step(b)
        ev[i + 1].record()
        torch.cuda.synchronize()
    print(f"trial {trial}: {K} keys x {occ} occurrences over a uniform background, us per step (forward + apply): " + " ".join(f"{ev[i].elapsed_time(ev[i + 1]) * 1e3:.0f}" for i in range(3)), flush=True)
assert t.status() == 0

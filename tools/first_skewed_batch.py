"""What the FIRST skewed batches of a stream cost: 20 uniform training steps (located forward + Adagrad apply, 256K keys, 100M-key dim-64 table), then Zipf(1.05)
batches — the time of each of the first 12, (a) with a host synchronisation per step (a training loop: the host sees batch i's skew report before it launches batch
i + 1) and (b) with the host running ahead (all 12 launched before the first one's report is back).  usage: first_skewed_batch.py [keys]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
keys_n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
batch, dim = 1 << 18, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
uni = bench.lookup_batches(synth, keys_n, batch, 8, "uniform", dev, seed=3)
zipf = bench.lookup_batches(synth, keys_n, batch, 12, "zipf", dev, seed=4)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev); slots = torch.empty(batch, dtype=torch.int64, device=dev)


def step(b):
    t.find_located(b, out=out, found=found, slots=slots, prepare_apply=True)
    t.apply_adagrad(b, grads, lr=0.01, slots=slots)


for mode in ("host_in_step", "host_runs_ahead"):
    for i in range(20):
        step(uni[i % 8])
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    ev[0].record()
    for i in range(12):
        step(zipf[i])
        ev[i + 1].record()
        if mode == "host_in_step":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(12)]
    print(f"{mode}: Zipf(1.05) steps behind a uniform stream, us per step (forward + apply): " + " ".join(f"{u:.0f}" for u in us), flush=True)
    if os.environ.get("MEE_FSB_SHORT"):   # under the profiler (tools/kernel_sequence.py DIR -30): the trace ends with these 12 steps
        break
    for i in range(70):   # back to a uniform stream long enough for the FULL kernel's stickiness (64 batches) to run out
        step(uni[i % 8])
    torch.cuda.synchronize()
assert t.status() == 0

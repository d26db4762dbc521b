"""find_or_insert at three miss rates (0 %, 5 %, 100 % new keys), 1M- and 256K-key batches, 100M-key table (MEE_LIB_PATH picks the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, INIT_UNIFORM, synth
dev = torch.device("cuda", 0)
N, dim = 100_000_000, 64
t = LookupTable(int(N / 0.70), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD, initializer=INIT_UNIFORM, init_scale=0.05)
bench.populate(t, synth, N, dim, dev, 1 << 20)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fresh_at = [N + (1 << 24)]
def fresh(n):
    k = synth.keys_t(1, fresh_at[0], n, dev); fresh_at[0] += n
    return k
for B in (1 << 20, 1 << 18):
    out = torch.empty((B, dim), dtype=torch.float32, device=dev); found = torch.empty(B, dtype=torch.uint8, device=dev)
    for name, frac in (("all present", 0.0), ("5 % new", 0.05), ("all new", 1.0)):
        reps = 6
        batches = []
        for r in range(reps + 1):
            k = synth.keys_t(1, (r * 7 + 3) * B, B, dev).clone()
            m = int(B * frac)
            if m:
                idx = torch.randperm(B, device=dev)[:m]
                k[idx] = fresh(m)
            batches.append(k)
        t.find_or_insert(batches[0], out=out, found=found)     # warm (its new keys are spent)
        torch.cuda.synchronize(); e0.record()
        for r in range(1, reps + 1):
            t.find_or_insert(batches[r], out=out, found=found)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"find_or_insert, {B} keys, {name}: {us:.0f} us -> {B / us / 1e3:.2f} G keys/s", flush=True)
assert t.status() == 0

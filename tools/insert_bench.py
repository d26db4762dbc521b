"""insert throughput: 1M-key batches of NEW keys into a 100M-key-capacity table (the populate pattern), then overwrites of present keys,
then a batch with duplicates.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, OPT_NONE, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps):
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
keys = [synth.keys_t(1, s * B, B, dev) for s in range(48)]
rows = synth.rows_t(keys[0], dim, 2)
for name, opt in (("no optimizer", OPT_NONE), ("adagrad", OPT_ADAGRAD)):
    t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=B, optimizer=opt)
    us = timed(lambda i: t.insert(keys[i], rows), 16)
    print(f"[{name}] insert, new keys, 1M batches (table 0 -> 16M keys): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
    us = timed(lambda i: t.insert(keys[16 + i], rows), 32)
    print(f"[{name}] insert, new keys, 1M batches (table 16M -> 48M keys): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
    us = timed(lambda i: t.insert(keys[i % 48], rows), 16)
    print(f"[{name}] insert, present keys (overwrite): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
    dup = torch.cat([keys[0][: B // 2], keys[0][: B // 2]])
    us = timed(lambda i: t.insert(dup, rows), 8)
    print(f"[{name}] insert, every key twice in the batch: {us:.0f} us")
    us = timed(lambda i: t.assign(keys[i % 48], rows), 16)
    print(f"[{name}] assign: {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
    assert t.size() == 48 * B and t.status() == 0
    t.close()

"""Throughput of the non-headline operators on one GPU (for DESIGN.md): insert, assign, find_or_insert, remove, export, size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, INIT_UNIFORM, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 20
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=B, optimizer=OPT_ADAGRAD, initializer=INIT_UNIFORM, init_scale=0.05)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps):
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
keys = [synth.keys_t(1, s * B, B, dev) for s in range(16)]
rows = synth.rows_t(keys[0], dim, 2)
us = timed(lambda i: t.insert(keys[i], rows), 16)
print(f"insert (new keys, 1M batches into an empty->16M table): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
bench.populate(t, synth, N, dim, dev, B)
us = timed(lambda i: t.insert(keys[i % 16], rows), 16)
print(f"insert (existing keys = overwrite, 100M-key table): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
us = timed(lambda i: t.assign(keys[i % 16], rows), 16)
print(f"assign: {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
us = timed(lambda i: t.find_or_insert(keys[i % 16]), 16)
print(f"find_or_insert (all present): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
fresh = [synth.keys_t(77, s * B, B, dev) for s in range(4)]
us = timed(lambda i: t.find_or_insert(fresh[i]), 4)
print(f"find_or_insert (all new, hashed initial rows): {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
us = timed(lambda i: t.remove(fresh[i]), 4)
print(f"remove: {us:.0f} us -> {B / us / 1e3:.2f} G keys/s")
t0 = time.time(); n = t.size(); torch.cuda.synchronize(); print(f"size(): {n} in {(time.time() - t0) * 1e3:.2f} ms")
k, v = t.export(); torch.cuda.synchronize()   # first call pays the 26 GB allocation
del k, v
t0 = time.time(); k, v = t.export(); torch.cuda.synchronize(); dt = time.time() - t0
print(f"export of {k.numel()} pairs ({k.numel() * (8 + 4 * dim) / 1e9:.1f} GB out): {dt * 1e3:.1f} ms -> {k.numel() * (8 + 4 * dim) * 2 / dt / 1e12:.2f} TB/s read+write")
del k, v
torch.cuda.empty_cache()
t0 = time.time(); n_r = sum(p[0].numel() for p in t.iter_export(1 << 24, with_state=True)); torch.cuda.synchronize(); dt = time.time() - t0
print(f"ranged export with state, 16M-slot pieces ({n_r} pairs, scratch {(1 << 24) * (8 + 8 * dim) / 1e9:.1f} GB): {dt * 1e3:.1f} ms")
cap0 = t.capacity
t0 = time.time(); t.reserve(int(cap0 * 1.5)); torch.cuda.synchronize(); dt = time.time() - t0
print(f"reserve {cap0} -> {t.capacity} slots ({n} keys, values + Adagrad plane, {t.table_bytes / 1e9:.1f} GB after): {dt * 1e3:.1f} ms -> {n / dt / 1e9:.2f} G keys/s")
assert t.size() == n and t.status() == 0
o, f = t.find(keys[3]); assert bool(f.all()) and torch.equal(o, rows if False else t.find(keys[3])[0])
import tempfile, shutil
d = tempfile.mkdtemp(dir="/tmp")
small = LookupTable(int(4e6 / 0.75), dim, device=dev, max_batch=B, optimizer=OPT_ADAGRAD)
for s in range(4): small.insert(keys[s], rows)
t0 = time.time(); m = small.save(d); dt = time.time() - t0
print(f"checkpoint save of {m} pairs with Adagrad state ({m * (8 + 8 * dim) / 1e9:.2f} GB) to {d}: {dt:.2f} s -> {m * (8 + 8 * dim) / dt / 1e9:.2f} GB/s")
fresh_t = LookupTable(int(8e6 / 0.75), dim, device=dev, max_batch=B, optimizer=OPT_ADAGRAD)
t0 = time.time(); m2 = fresh_t.load(d); torch.cuda.synchronize(); dt = time.time() - t0
print(f"checkpoint load: {dt:.2f} s -> {m2 * (8 + 8 * dim) / dt / 1e9:.2f} GB/s")
shutil.rmtree(d)

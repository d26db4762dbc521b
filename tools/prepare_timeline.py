"""Roles inside the training forward's launch (diagnostic build: tools/build_variant.sh ftl -DMEE_FIND_TIMELINE=1; MEE_LIB_PATH=build/libmeepo_hip_ftl.so): when the
partition blocks end, when the find blocks end.   usage: prepare_timeline.py [uniform|zipf]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth, _lib
dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
batch, keys_n, dim = 1 << 18, 100_000_000, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev); slots = torch.empty(batch, dtype=torch.int64, device=dev)
L = _lib.lib()
L.mee_debug_find_timeline.argtypes = [C.c_void_p, C.c_uint64]; L.mee_debug_find_timeline.restype = C.c_int
def step(i):
    t.find_located(batches[i % 8], out=out, found=found, slots=slots, prepare_apply=True)
    t.apply_adagrad(batches[i % 8], grads, lr=0.01, slots=slots)
for i in range(12):
    step(i); torch.cuda.synchronize()
assert L.mee_debug_find_timeline(None, 0) == 0   # arm
for i in range(12, 20):
    step(i)
torch.cuda.synchronize()
buf = np.zeros(16384 * 4, dtype=np.uint64)
assert L.mee_debug_find_timeline(buf.ctypes.data, buf.size) == 0
r = buf.reshape(16384, 4)
r = r[r[:, 0] > 0]
t0 = r[:, 0].min()
start = (r[:, 0] - t0).astype(np.float64) * 0.01; end = (r[:, 1] - t0).astype(np.float64) * 0.01; role = r[:, 2].astype(np.int64)
p, f = role == 100, role == 1
print(f"{dist}: {p.sum()} partition blocks: start median {np.median(start[p]):.2f} us, end median {np.median(end[p]):.2f}, p90 {np.percentile(end[p], 90):.2f}, max {end[p].max():.2f}")
print(f"{dist}: {f.sum()} find blocks (the first 16384 of the grid): last start {start[f].max():.2f} us, end median {np.median(end[f]):.2f}, max {end[f].max():.2f}; block life median {np.median((end - start)[f]):.2f}")

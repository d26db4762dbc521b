"""Phase timeline of the bucketed apply's bucket blocks (diagnostic build: tools/build_variant.sh tl -DMEE_APPLY_TIMELINE=1; run with
MEE_LIB_PATH=build/libmeepo_hip_tl.so).  Stamps per block (100 MHz wall clock): 0 entry, 1 first round trip done, 2 entries fetched + keys in the
LDS table, 3 scans done, 4 thread 0's items done, 5 block done.   usage: apply_timeline.py [uniform|zipf] [located 0|1]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth, _lib
dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
located = int(sys.argv[2]) if len(sys.argv) > 2 else 1
keys_n, batch, dim = 100_000_000, 1 << 18, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev); slots = torch.empty(batch, dtype=torch.int64, device=dev)
for i in range(12):
    if located:
        t.find_located(batches[i % 8], out=out, found=found, slots=slots, prepare_apply=True)
        t.apply_adagrad(batches[i % 8], grads, lr=0.01, slots=slots)
    else:
        t.apply_adagrad(batches[i % 8], grads, lr=0.01)
torch.cuda.synchronize()
n_blocks = 256 + 768
buf = np.zeros(n_blocks * 8, dtype=np.uint64)
L = _lib.lib()
L.mee_debug_timeline.argtypes = [C.c_void_p, C.c_uint64]; L.mee_debug_timeline.restype = C.c_int
assert L.mee_debug_timeline(buf.ctypes.data, buf.size) == 0
tl = buf.reshape(n_blocks, 8)[256:, :6].astype(np.float64) * 0.01   # us; bucket blocks only
t0 = tl[:, 0].min()
print(f"{dist}, located={located}: {tl.shape[0]} bucket blocks; block start relative to the first: median {np.median(tl[:, 0] - t0):.2f} us, p90 {np.percentile(tl[:, 0] - t0, 90):.2f}, max {(tl[:, 0] - t0).max():.2f}")
names = ["first round trip (selector, totals, run matrix)", "entries + keys into the LDS table", "scans + source sort (+ slot handles)", "work items (thread 0)", "rest of the block"]
for k in range(5):
    d = tl[:, k + 1] - tl[:, k]
    print(f"  {names[k]:52s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}   max {d.max():6.2f}")
end = tl[:, 5] - t0
print(f"  block end relative to the first start: median {np.median(end):.2f} us, p90 {np.percentile(end, 90):.2f}, max {end.max():.2f}")

raw = buf.reshape(n_blocks, 8)[256:]
bucket = (raw[:, 6] >> np.uint64(32)).astype(np.int64); size = (raw[:, 6] & np.uint64(0xffffffff)).astype(np.float64); xcc = (raw[:, 7] >> np.uint64(32)).astype(np.int64); hw = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7    # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
items = tl[:, 4] - tl[:, 3]
print(f"  bucket sizes: median {np.median(size):.0f}, p10 {np.percentile(size, 10):.0f}, p90 {np.percentile(size, 90):.0f}; corr(items time, size) = {np.corrcoef(items, size)[0, 1]:.2f}; corr(items time, block index) = {np.corrcoef(items, np.arange(items.size))[0, 1]:.2f}")
print("  items time per position: median %.1f ns, p10 %.1f, p90 %.1f" % tuple(np.percentile(items / size * 1e3, [50, 10, 90])))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: {m.sum():4d} blocks, items median {np.median(items[m]):6.2f} us, end median {np.median(end[m]):6.2f}, end max {end[m].max():6.2f}")
key = xcc * 1000 + se * 100 + sh * 20 + cu
cnt = {k: int((key == k).sum()) for k in set(key.tolist())}
per = np.array([cnt[k] for k in key.tolist()])
for c in sorted(set(per.tolist())):
    m = per == c
    print(f"  blocks on a CU that holds {c} bucket blocks: {m.sum():4d}, items median {np.median(items[m]):6.2f} us, end median {np.median(end[m]):6.2f}")

for name, par in (("XCC parity", xcc & 1), ("bucket parity", bucket & 1)):
    print(f"  by {name}: even items median {np.median(items[par == 0]):6.2f} us, odd {np.median(items[par == 1]):6.2f} us")
q = np.argsort(bucket)
print("  items time by bucket index (tenths of the bucket range): " + " ".join(f"{np.median(items[q][i * len(q) // 10:(i + 1) * len(q) // 10]):.1f}" for i in range(10)))

for k in range(4):
    d = tl[:, k + 1] - tl[:, k]
    print(f"  phase {k} by XCC parity: even {np.median(d[(xcc & 1) == 0]):6.2f} us, odd {np.median(d[(xcc & 1) == 1]):6.2f} us")

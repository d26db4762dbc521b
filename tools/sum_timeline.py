"""Phase timeline of mee_dedup_sum's blocks (diagnostic build: MEE_VARIANT_TU=meepo_dedup tools/build_variant.sh stl -DMEE_SUM_TIMELINE=1; run with
MEE_LIB_PATH=build/libmeepo_hip_stl.so).  Stamps per block (100 MHz wall clock): 0 entry, 1 totals / runs / prefix in, 2 entries fetched + keys in the LDS
table, 3 scan done + keys and counts written, 4 look-ups + inverse + sorted source list, 5 long and medium runs, 6 short runs (thread 0).
usage: sum_timeline.py [uniform|zipf] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, synth, _lib
dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
keys_n, dim = 100_000_000, 64
t = LookupTable(64, dim, device=dev, max_batch=batch)
bs = bench.lookup_batches(synth, keys_n, batch, 4, dist, dev, seed=3)
rows = torch.randn(batch, dim, device=dev)
L = _lib.lib()
L.mee_debug_sum_timeline.argtypes = [C.c_void_p, C.c_uint64]; L.mee_debug_sum_timeline.restype = C.c_int
drop = os.environ.get("MEE_STL_NULL", "").split(",")   # outputs left out (inverse, counts): what their stores cost
uo = torch.empty(batch, dtype=torch.int64, device=dev); go = torch.empty((batch, dim), device=dev)
co = torch.empty(batch, dtype=torch.int32, device=dev); io = torch.empty(batch, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def call(b):
    _lib.check(L.mee_dedup_sum(t._h, b.data_ptr(), rows.data_ptr(), batch, uo.data_ptr(), go.data_ptr(), None if "counts" in drop else co.data_ptr(),
                               None if "inverse" in drop else io.data_ptr(), -1, st))
for i in range(6):
    call(bs[i % 4])
    torch.cuda.synchronize()
assert L.mee_debug_sum_timeline(None, 0) == 0          # arm
call(bs[2])
torch.cuda.synchronize()
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert L.mee_debug_sum_timeline(buf.ctypes.data, buf.size) == 0
tl = buf.reshape(8192, 16)
used = tl[:, 0] != 0
size = (tl[:, 15] & 0xFFFFFFFF).astype(np.int64); window = (tl[:, 15] >> 32) != 0
us = tl[:, :12].astype(np.float64) * 0.01
t0 = us[used, 0].min()
print(f"{dist}, {batch} keys: {int(used.sum())} blocks stamped, {int((used & window).sum())} of them windows of hot keys' buckets; sizes: median {np.median(size[used & ~window & (size > 0)]):.0f}, max {size[used].max()}")
print(f"block start relative to the first: median {np.median(us[used, 0] - t0):.1f} us, p90 {np.percentile(us[used, 0] - t0, 90):.1f}, max {(us[used, 0] - t0).max():.1f}")
names = ["totals / runs / prefix (first round trip)", "entries fetched + keys into the LDS table", "scan + keys and counts written", "look-ups + inverse + sorted source list", "long + medium runs", "short runs (thread 0's tile)"]
hb = used & ~window & (us[:, 6] > 0)
for k in range(6):
    d = us[hb, k + 1] - us[hb, k]
    print(f"  {names[k]:46s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}")
for a, b_, nm in ((1, 7, "  .. table cleared"), (7, 8, "  .. thread 0's entries in (binary search + one load)"), (8, 2, "  .. keys into the table + barrier"), (3, 9, "  .. thread 0's look-ups + inverse stores issued"), (9, 4, "  .. barrier, list filed, barrier")):
    d = us[hb, b_] - us[hb, a]
    print(f"  {nm:58s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
life = us[hb, 6] - us[hb, 0]
print(f"  hash-bucket block life: median {np.median(life):.1f} us, p90 {np.percentile(life, 90):.1f}, max {life.max():.1f}; last end {us[hb, 6].max() - t0:.1f} us after the first start")
lf = np.where(hb, us[:, 6] - us[:, 0], 0.0)
for bi in np.argsort(lf)[-3:][::-1]:
    print(f"  slowest: block {bi} size {size[bi]} start {us[bi, 0] - t0:.1f} phases " + " ".join(f"{us[bi, k + 1] - us[bi, k]:.1f}" for k in range(6)))
rowp = us[:, 6] - us[:, 5]
idx = np.arange(8192)
print("  short-run phase by blockIdx % 8 (XCD): " + "  ".join(f"{np.median(rowp[hb & (idx % 8 == x)]):.0f}" for x in range(8)) + " us;  block end by XCD: "
      + "  ".join(f"{np.median(us[hb & (idx % 8 == x), 6] - t0):.0f}/{(us[hb & (idx % 8 == x), 6] - t0).max():.0f}" for x in range(8)))
wb = used & window
if wb.any() and (us[wb, 10] > 0).any():
    real = wb & (size > 0)
    for nm, m in (("blocks with a window", real), ("blocks without one", wb & ~real)):
        if m.any():
            print(f"  {nm}: {int(m.sum())}; start median {np.median(us[m, 0] - t0):.1f} us (p90 {np.percentile(us[m, 0] - t0, 90):.1f}); window done median {np.median(us[m, 10] - t0):.1f} (max {(us[m, 10] - t0).max():.1f}); "
                  f"long runs of the hash buckets done median {np.median(us[m, 11] - t0):.1f} (max {(us[m, 11] - t0).max():.1f})")
if wb.any():
    print(f"  window blocks: start median {np.median(us[wb, 0] - t0):.1f} us, first phase {np.median(us[wb, 1] - us[wb, 0]):.1f} us (their later phases are not stamped)")

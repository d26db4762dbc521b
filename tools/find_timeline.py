"""Per-XCD timing of the find kernel's blocks (diagnostic build: tools/build_variant.sh ftl -DMEE_FIND_TIMELINE=1; MEE_LIB_PATH=build/libmeepo_hip_ftl.so).
usage: find_timeline.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, synth, _lib
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
keys_n, dim = 100_000_000, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, "uniform", dev, seed=3)
outs = [torch.empty((batch, dim), device=dev) for _ in range(6)]; found = torch.empty(batch, dtype=torch.uint8, device=dev)
L = _lib.lib()
L.mee_debug_find_timeline.argtypes = [C.c_void_p, C.c_uint64]; L.mee_debug_find_timeline.restype = C.c_int
for nt in (-1, 1):
    t.set_tuning("find_nt", nt)
    for i in range(20):
        t.find(batches[i % 8], out=outs[i % 6], found=found)
    torch.cuda.synchronize()
    if nt == -1:
        assert L.mee_debug_find_timeline(None, 0) == 0   # arm
        continue
    n_blocks = min(16384, batch // 32)
    buf = np.zeros(16384 * 4, dtype=np.uint64)
    assert L.mee_debug_find_timeline(buf.ctypes.data, buf.size) == 0
    r = buf.reshape(16384, 4)[:n_blocks]
    t0 = r[:, 0].min()
    start = (r[:, 0] - t0).astype(np.float64) * 0.01; end = (r[:, 1] - t0).astype(np.float64) * 0.01; xcc = r[:, 2].astype(np.int64)
    print(f"find of {batch} keys, find_nt={nt}: {n_blocks} blocks; last block ends at {end.max():.2f} us; block life median {np.median(end - start):.2f} us")
    for x in range(8):
        m = xcc == x
        print(f"  XCC {x}: {m.sum():5d} blocks, last start {start[m].max():6.2f} us, last end {end[m].max():6.2f} us, median block life {np.median((end - start)[m]):5.2f} us")

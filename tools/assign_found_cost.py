"""mee_assign per 1M uniform keys with and without the found bytes (d_found = NULL): what the 1M scattered byte stores cost.  usage: assign_found_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth, _lib
from meepoembedding_amd._lib import check
dev = torch.device("cuda", 0)
keys_n, batch, dim = 100_000_000, 1 << 20, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=batch)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
bs = bench.lookup_batches(synth, keys_n, batch, 8, "uniform", dev, seed=3)
rows = torch.randn(batch, dim, device=dev)
found = torch.empty(batch, dtype=torch.uint8, device=dev)
L = _lib.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream(dev).cuda_stream
for label, fp in (("with found bytes", found.data_ptr()), ("d_found = NULL", 0), ("with found bytes", found.data_ptr()), ("d_found = NULL", 0)):
    for i in range(5):
        check(L.mee_assign(t._h, bs[i % 8].data_ptr(), rows.data_ptr(), batch, fp, st))
    torch.cuda.synchronize()
    e0.record()
    for i in range(30):
        check(L.mee_assign(t._h, bs[i % 8].data_ptr(), rows.data_ptr(), batch, fp, st))
    e1.record(); torch.cuda.synchronize()
    print(f"assign, 1M uniform keys, {label}: {e0.elapsed_time(e1) * 1e3 / 30:.1f} us", flush=True)

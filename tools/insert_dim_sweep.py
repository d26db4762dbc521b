import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
B = 1 << 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for dim in (4, 16, 64, 128):
    t = LookupTable(int(60e6 / 0.75), dim, device=dev, max_batch=B)
    keys = [synth.keys_t(1, s * B, B, dev) for s in range(40)]
    rows = synth.rows_t(keys[0], dim, 2)
    for i in range(8): t.insert(keys[i], rows)
    torch.cuda.synchronize(); e0.record()
    for i in range(8, 40): t.insert(keys[i], rows)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 32
    print(f"dim {dim}: insert 1M new keys {us:.0f} us -> {B / us / 1e3:.2f} G keys/s, {(8 + 128 + 8 * dim) * B / us / 1e6:.2f} TB/s of key+line+row read+row write")
    t.close()

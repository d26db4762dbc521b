import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
t = LookupTable(1 << 16, 64, device=dev, max_batch=1 << 16)
k = synth.keys_t(1, 0, 64, dev)
out = torch.empty((64, 64), device=dev); found = torch.empty(64, dtype=torch.uint8, device=dev)
for _ in range(1000): t.find(k, out=out, found=found)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): t.find(k, out=out, found=found)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host cost per find() call: {(t1 - t0) / 20000 * 1e6:.2f} us (GPU side ~8 us per tiny launch)")

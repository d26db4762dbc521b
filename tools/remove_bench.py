"""remove / locate at 1M keys on a 100M-key table (MEE_LIB_PATH picks the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
N, dim, B = 100_000_000, 64, 1 << 20
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=B)
bench.populate(t, synth, N, dim, dev, B)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ks = [synth.keys_t(1, s * B, B, dev) for s in range(12)]
for i in range(2): t.locate(ks[i])
torch.cuda.synchronize(); e0.record()
for i in range(10): t.locate(ks[i])
e1.record(); torch.cuda.synchronize(); print(f"locate: {e0.elapsed_time(e1) * 100:.1f} us per 1M keys")
torch.cuda.synchronize(); e0.record()
for i in range(8): t.remove(ks[i])
e1.record(); torch.cuda.synchronize(); print(f"remove: {e0.elapsed_time(e1) * 125:.1f} us per 1M keys")

"""One launch for many tables vs one launch per table (mee_find_grouped vs mee_find): a recsys-shaped collection of
T tables x K keys each, small per-table batches (where the ~7.6 us launch floor dominates a per-table loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, TableGroup, synth
dev = torch.device("cuda", 0)
T_, K, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 26, 4_000_000, 64
tables = []
for j in range(T_):
    t = LookupTable(int(K / 0.75), dim, device=dev, max_batch=1 << 20)
    for s in range(0, K, 1 << 20):
        k = synth.keys_t(100 + j, s, min(1 << 20, K - s), dev)
        t.insert(k, synth.rows_t(k, dim, 2))
    tables.append(t)
grp = TableGroup(tables)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f"{T_} tables x {K} keys, dim {dim}, {sum(t.table_bytes for t in tables) / 1e9:.1f} GB")
for per in (512, 2048, 8192, 32768):
    n = per * T_
    gen = torch.Generator(device="cpu").manual_seed(per)
    batches = [[synth.keys_t(100 + j, 0, K, dev)[torch.randint(0, K, (per,), generator=gen).to(dev)] for j in range(T_)] for _ in range(4)]
    cat = [torch.cat(b) for b in batches]
    offs = torch.arange(0, n + 1, per, dtype=torch.int64, device=dev)
    out = torch.empty((n, dim), device=dev); found = torch.empty(n, dtype=torch.uint8, device=dev)
    def looped(i):
        for j, t in enumerate(tables):
            t.find(batches[i % 4][j], out=out[j * per:(j + 1) * per], found=found[j * per:(j + 1) * per])
    def grouped(i):
        grp.find(cat[i % 4], offs, out=out, found=found)
    res = {}
    for name, fn in (("looped", looped), ("grouped", grouped)):
        for i in range(5): fn(i)
        torch.cuda.synchronize(); e0.record()
        for i in range(50): fn(i)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 50
        chk = out.clone()
        res[name + "_out"] = chk
    assert torch.equal(res["looped_out"], res["grouped_out"]) and bool(found.all())
    print(f"{per:6d} keys/table ({n} per step): per-table loop {res['looped']:.1f} us, grouped {res['grouped']:.1f} us "
          f"({res['looped'] / res['grouped']:.1f}x; {n / res['grouped'] / 1e3:.2f} G lookups/s, {n * 528 / res['grouped'] / 1e3 / 8000:.2f} of the HBM roofline)")

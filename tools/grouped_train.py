"""Train step of an embedding collection: grouped (2 calls; the group's apply through the group table — 7 launches — and through the bucketed
path — partition + one update kernel) vs a per-table loop (2 calls per table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, TableGroup, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
T_, K, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 26, int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000, 64
PERS = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (512, 2048, 8192, 32768)
tables = []
for j in range(T_):
    t = LookupTable(int(K / 0.75), dim, device=dev, max_batch=max(1 << 16, max(PERS)), optimizer=OPT_ADAGRAD)
    for s in range(0, K, 1 << 16):
        k = synth.keys_t(100 + j, s, min(1 << 16, K - s), dev)
        t.insert(k, synth.rows_t(k, dim, 2))
    tables.append(t)
print(f"{T_} tables x {K} keys, dim {dim}, Adagrad, {sum(t.table_bytes for t in tables) / 1e9:.1f} GB")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for per in PERS:
    n = per * T_
    grp = TableGroup(tables, max_apply_batch=n)
    gen = torch.Generator(device="cpu").manual_seed(per)
    batches = [[synth.keys_t(100 + j, 0, K, dev)[torch.randint(0, K, (per,), generator=gen).to(dev)] for j in range(T_)] for _ in range(4)]
    cat = [torch.cat(b) for b in batches]
    grads = torch.randn(n, dim, device=dev) * 0.01
    offs = torch.arange(0, n + 1, per, dtype=torch.int64, device=dev)
    out = torch.empty((n, dim), device=dev); found = torch.empty(n, dtype=torch.uint8, device=dev)
    def looped(i):
        for j, t in enumerate(tables):
            t.find(batches[i % 4][j], out=out[j * per:(j + 1) * per], found=found[j * per:(j + 1) * per])
        for j, t in enumerate(tables):
            t.apply_adagrad(batches[i % 4][j], grads[j * per:(j + 1) * per], lr=0.01)
    def grouped(i):
        grp.find(cat[i % 4], offs, out=out, found=found)
        grp.apply_adagrad(cat[i % 4], offs, grads, lr=0.01)
    res = {}
    for name, fn in (("looped", looped), ("grouped", grouped)):
        for i in range(3): fn(i)
        torch.cuda.synchronize(); e0.record()
        for i in range(20): fn(i)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 20
    uniq = sum(int(torch.unique(c).numel()) for c in cat) / 4
    bytes_ = (528 + 264) * n + 1032 * uniq
    print(f"{per:6d} keys/table ({n} per step): per-table loop {res['looped']:.0f} us, grouped {res['grouped']:.1f} us "
          f"({res['looped'] / res['grouped']:.1f}x; {n / res['grouped'] / 1e3:.2f} G keys/s, {bytes_ / res['grouped'] / 1e3 / 8000:.2f} of the HBM roofline)", flush=True)
    grp.close()

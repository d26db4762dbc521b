"""An embedding-bag collection step (torchrec-EBC shape): T tables, B samples, L ids per bag.  One pooled launch + one grouped
optimizer step vs a per-table loop of pooled lookups + indexed applies vs the unfused per-table find + index_add."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meepoembedding_amd import LookupTable, TableGroup, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
T_, K, dim, L = 26, 4_000_000, 64, 5
tables = []
for j in range(T_):
    t = LookupTable(int(K / 0.75), dim, device=dev, max_batch=1 << 17, optimizer=OPT_ADAGRAD)
    for s in range(0, K, 1 << 17):
        k = synth.keys_t(100 + j, s, min(1 << 17, K - s), dev)
        t.insert(k, synth.rows_t(k, dim, 2))
    tables.append(t)
print(f"{T_} tables x {K} keys, dim {dim}, Adagrad, bags of {L}", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
BS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (512, 2048, 8192)
for B in BS:
    n = T_ * B * L
    grp = TableGroup(tables, max_apply_batch=n)
    gen = torch.Generator(device="cpu").manual_seed(B)
    segs = [synth.keys_t(100 + j, 0, K, dev)[torch.randint(0, K, (B * L,), generator=gen).to(dev)] for j in range(T_)]
    keys = torch.cat(segs)
    off = torch.arange(0, n + 1, L, dtype=torch.int64, device=dev)
    off1 = torch.arange(0, B * L + 1, L, dtype=torch.int64, device=dev)
    bag_of = torch.repeat_interleave(torch.arange(T_ * B, device=dev), L)
    bag_of1 = torch.repeat_interleave(torch.arange(B, device=dev), L)
    bag_grads = torch.randn(T_ * B, dim, device=dev) * 0.01
    out = torch.empty((T_ * B, dim), device=dev); found = torch.empty(n, dtype=torch.uint8, device=dev)
    rows = torch.empty((B * L, dim), device=dev)
    located = torch.empty(n, dtype=torch.int64, device=dev)
    def grouped(i):
        grp.find_pooled(keys, off, "sum", out=out, found=found, located=located)
        grp.apply_pooled(keys, off, bag_grads, bag_of, "adagrad", lr=0.01, located=located)
    def looped(i):
        for j, t in enumerate(tables):
            t.find_pooled(segs[j], off1, "sum", out=out[j * B:(j + 1) * B], found=found[j * B * L:(j + 1) * B * L])
        for j, t in enumerate(tables):
            t.apply_adagrad(segs[j], bag_grads[j * B:(j + 1) * B], lr=0.01, grad_index=bag_of1)
    def unfused(i):
        for j, t in enumerate(tables):
            t.find(segs[j], out=rows, found=found[j * B * L:(j + 1) * B * L])
            o = out[j * B:(j + 1) * B]; o.zero_(); o.index_add_(0, bag_of1, rows)
        for j, t in enumerate(tables):
            t.apply_adagrad(segs[j], bag_grads[j * B:(j + 1) * B][bag_of1], lr=0.01)
    tg, tl, tu = timed(grouped), timed(looped), timed(unfused)
    print(f"batch {B:5d} ({n} ids per step): unfused per-table {tu:.0f} us, pooled per-table {tl:.0f} us, grouped + pooled {tg:.1f} us "
          f"({tu / tg:.1f}x / {tl / tg:.1f}x; {n / tg / 1e3:.2f} G ids/s)", flush=True)
    grp.close()

"""GPU timing harness for the sparse-optimizer path (configs[2]): find + apply_adagrad/adam per step.
usage: python tools/tune_apply.py [--keys N] [--batch B] [--opt adagrad|adam] [--dist uniform|zipf] [--launches L]"""
import argparse, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, OPT_ADAM, synth

ap = argparse.ArgumentParser()
ap.add_argument("--keys", type=int, default=100_000_000)
ap.add_argument("--batch", type=int, default=1 << 18)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--opt", default="adagrad")
ap.add_argument("--dist", default="uniform")
ap.add_argument("--launches", type=int, default=100)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--apply-rounds", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
kind = OPT_ADAGRAD if a.opt == "adagrad" else OPT_ADAM
t = LookupTable(int(a.keys / 0.75), a.dim, device=dev, max_batch=1 << 20, optimizer=kind)
bench.populate(t, synth, a.keys, a.dim, dev, 1 << 20)
NB = 16
batches = bench.lookup_batches(synth, a.keys, a.batch, NB, a.dist, dev, seed=3)
uniq = [int(torch.unique(b).numel()) for b in batches]
grads = [torch.randn(a.batch, a.dim, device=dev) * 0.01 for _ in range(4)]
out = torch.empty((a.batch, a.dim), dtype=torch.float32, device=dev); found = torch.empty(a.batch, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def apply(i):
    if kind == OPT_ADAGRAD: t.apply_adagrad(batches[i % NB], grads[i % 4], lr=0.01, eps=1e-10)
    else: t.apply_adam(batches[i % NB], grads[i % 4], lr=0.001, step=i + 1)

def timeit(fn):
    ts = []
    for r in range(a.rounds):
        for i in range(5): fn(i)
        torch.cuda.synchronize(); e0.record()
        for i in range(a.launches): fn(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / a.launches)
    return statistics.median(ts)

U = statistics.mean(uniq); B = a.batch
row = 4 * a.dim
per_u = 8 + (4 if kind == OPT_ADAGRAD else 6) * row   # table key + w, state read + written (1032 / 1544 B at dim 64)
ab = (8 + row) * B + per_u * U
slots = torch.empty(a.batch, dtype=torch.int64, device=dev)

def located_step(i):
    t.find_located(batches[i % NB], out=out, found=found, slots=slots)
    if kind == OPT_ADAGRAD: t.apply_adagrad(batches[i % NB], grads[i % 4], lr=0.01, eps=1e-10, slots=slots)
    else: t.apply_adam(batches[i % NB], grads[i % 4], lr=0.001, step=i + 1, slots=slots)

def located_apply(i):
    if kind == OPT_ADAGRAD: t.apply_adagrad(batches[i % NB], grads[i % 4], lr=0.01, eps=1e-10, slots=slots)
    else: t.apply_adam(batches[i % NB], grads[i % 4], lr=0.001, step=i + 1, slots=slots)

for nt_ in (-1, 0, 1, 4, 5):   # cache policy of the forward find INSIDE the training step (the apply's traffic sweeps the Infinity Cache between finds)
    t.set_tuning('find_nt', nt_)
    print(f"find_nt={nt_}: find_located + located apply step {timeit(located_step):.1f} us, plain find + apply step {timeit(lambda i: (t.find(batches[i % NB], out=out, found=found), apply(i))):.1f} us")
t.set_tuning('find_nt', -1)
for ov in (0, 1):
    ta = timeit(apply)
    t.find_located(batches[0], out=out, found=found, slots=slots)
    tl = timeit(lambda i: (t.apply_adagrad(batches[0], grads[i % 4], lr=0.01, eps=1e-10, slots=slots) if kind == OPT_ADAGRAD else
                           t.apply_adam(batches[0], grads[i % 4], lr=0.001, step=i + 1, slots=slots)))
    ts = timeit(located_step)
    print(f"round {ov}: apply {ta:.1f} us ({ab / ta / 1e3 / 8000:.3f}), located apply (one batch) {tl:.1f} us, find_located + located apply step {ts:.1f} us "
          f"({((16 + 2 * row) * B + ab) / ts / 1e3 / 8000:.3f} of the step roofline)")
t_apply = timeit(apply)
t_find = timeit(lambda i: t.find(batches[i % NB], out=out, found=found))
t_step = timeit(lambda i: (t.find(batches[i % NB], out=out, found=found), apply(i)))
print(f"{a.opt} {a.dist} batch {B} unique {U:.0f}: apply {t_apply:.1f} us ({ab / t_apply / 1e3:.0f} GB/s algorithmic = {ab / t_apply / 1e3 / 8000:.3f}), "
      f"find {t_find:.1f} us, find+apply step {t_step:.1f} us -> {B / t_step / 1e3:.2f} Gkeys/s ({((16 + 2 * row) * B + ab) / t_step / 1e3 / 8000:.3f} of roofline)")

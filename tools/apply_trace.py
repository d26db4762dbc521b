"""applies under rocprofv3 --kernel-trace / --pmc: per-kernel times of the apply alone and of the find_located + apply step.
usage: apply_trace.py [keys] [uniform|zipf] [ignored: round 3 took a list of apply paths here] [adagrad|adam]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, OPT_ADAM, synth
dev = torch.device("cuda", 0)
keys_n, batch, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 1 << 18, 64
dist_name = sys.argv[2] if len(sys.argv) > 2 else "uniform"
adam = len(sys.argv) > 4 and sys.argv[4] == "adam"
bmax = int(os.environ.get("MEE_BUCKET_MAX", "0"))
pdbg = int(os.environ.get("MEE_PREPARE_DEBUG", "0"))
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAM if adam else OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist_name, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), dtype=torch.float32, device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev)
slots = torch.empty(batch, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def apply(k, **kw):
    if adam:
        t.apply_adam(k, grads, lr=0.001, step=3, **kw)
    else:
        t.apply_adagrad(k, grads, lr=0.01, **kw)


for _ in (0,):
    if bmax:
        t.set_tuning("apply_bucket_max", bmax)
    if pdbg:
        t.set_tuning("prepare_debug", pdbg)
    if os.environ.get("MEE_XCD_SPLIT"):
        t.set_tuning("apply_xcd_split", int(os.environ["MEE_XCD_SPLIT"]))
    for label, located in (("apply alone (probing)", False), ("find_located + apply (located)", True), ("find_located_prepare + apply (located)", 2)):
        def step(i):
            if located:
                t.find_located(batches[i % 8], out=out, found=found, slots=slots, prepare_apply=located == 2)
                apply(batches[i % 8], slots=slots)
            else:
                apply(batches[i % 8])
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(50):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        print(f"apply_path 1 {dist_name} {'adam' if adam else 'adagrad'}: {label}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per step", flush=True)
assert t.status() == 0

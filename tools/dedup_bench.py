"""mee_dedup_keys and mee_assign per batch of 1M keys (uniform, Zipf 1.05), the bucketed machinery against round 2's group table
(tuning "dedup_path" = 0); sharded lookup with pre-exchange dedup at world 1 is bench.py --force-sharded --dist zipf --dedup.
usage: dedup_bench.py [keys] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, synth
dev = torch.device("cuda", 0)
keys_n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dim = 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=batch)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rows = torch.randn(batch, dim, device=dev)


def timeit(fn, reps=30):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]


for dist in os.environ.get("MEE_DEDUP_DIST", "uniform,zipf").split(","):
    bs = bench.lookup_batches(synth, keys_n, batch, 8, dist, dev, seed=3)
    for path, label in (((-1, "bucketed"),) if os.environ.get("MEE_DEDUP_ONLY_NEW") else ((-1, "bucketed"), (0, "group table (round 2)"))):
        t.set_tuning("dedup_path", path)
        u, inv = t.dedup_keys(bs[0])
        assert torch.equal(u[inv], bs[0]), "dedup_keys: uniq[inverse] != keys"
        nu = int((u != -(1 << 63)).sum())  # (EMPTY anywhere is padding)
        assert nu == int(torch.unique(bs[0]).numel())
        td = timeit(lambda i: t.dedup_keys(bs[i % 8]))
        print(f"{dist:8s} {label:24s}: dedup_keys {td:7.1f} us per {batch} keys ({nu} distinct)", flush=True)
        ta = timeit(lambda i: t.assign(bs[i % 8], rows))
        print(f"{dist:8s} {label:24s}: assign     {ta:7.1f} us per {batch} keys", flush=True)
t.set_tuning("dedup_path", -1)
assert t.status() == 0

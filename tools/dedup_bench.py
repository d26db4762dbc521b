"""mee_dedup_keys, mee_dedup_sum (dim-64 rows) and mee_assign per batch of 1M keys (uniform, Zipf 1.05); sharded lookup / training step with
pre-exchange dedup at world 1 is bench.py --force-sharded --dist zipf --dedup.  (MEE_LIB_PATH = an earlier round's library to compare with: round 4's
mee_dedup_sum was the group table's, synchronous — this script only times what the loaded library has.)
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
    u, inv = t.dedup_keys(bs[0])
    assert torch.equal(u[inv], bs[0]), "dedup_keys: uniq[inverse] != keys"
    nu = int((u != -(1 << 63)).sum())  # (EMPTY anywhere is padding)
    assert nu == int(torch.unique(bs[0]).numel())
    td = timeit(lambda i: t.dedup_keys(bs[i % 8]))
    print(f"{dist:8s}: dedup_keys {td:7.1f} us per {batch} keys ({nu} distinct)", flush=True)
    # dedup_sum: outputs preallocated once (the wrapper's torch.empty calls are not what is measured)
    import ctypes as C
    from meepoembedding_amd import _lib
    uo = torch.empty(batch, dtype=torch.int64, device=dev); go = torch.empty((batch, dim), device=dev)
    co = torch.empty(batch, dtype=torch.int32, device=dev); io = torch.empty(batch, dtype=torch.int64, device=dev)
    L, st = _lib.lib(), torch.cuda.current_stream(dev).cuda_stream
    def ds(i, with_rows=True):
        _lib.check(L.mee_dedup_sum(t._h, bs[i % 8].data_ptr(), rows.data_ptr() if with_rows else None, batch, uo.data_ptr(), go.data_ptr() if with_rows else None,
                                   co.data_ptr(), io.data_ptr(), -1, st))
    ds(0)
    torch.cuda.synchronize()
    keep = co > 0
    assert int(keep.sum()) == nu and torch.equal(uo[io], bs[0]) and int(co.sum()) == batch
    ref = torch.zeros((batch, dim), device=dev, dtype=torch.float64).index_add_(0, io, rows.double())
    assert torch.allclose(go[keep].double(), ref[keep], rtol=1e-6, atol=1e-9), "dedup_sum: summed rows"
    ts = timeit(ds)
    alg = batch * (8 + 4 * dim) + nu * (8 + 4 * dim + 4) + batch * 8
    print(f"{dist:8s}: dedup_sum  {ts:7.1f} us per {batch} keys with dim-{dim} rows ({alg / ts / 1e6:.2f} TB/s of its {alg / 1e6:.0f} MB algorithmic)", flush=True)
    tk = timeit(lambda i: ds(i, False))
    print(f"{dist:8s}: dedup_sum without rows (keys, counts, inverse) {tk:7.1f} us", flush=True)
    ta = timeit(lambda i: t.assign(bs[i % 8], rows))
    print(f"{dist:8s}: assign     {ta:7.1f} us per {batch} keys", flush=True)
assert t.status() == 0

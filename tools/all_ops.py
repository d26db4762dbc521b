"""Every operator of SURVEY 8a on one table, a few calls each, for per-kernel rocprofv3 passes (tools/pmc_all.sh): 100M keys, dim 64,
Adagrad plane, 1M-key batches for the mutators / partition, 256K-key batches for find and apply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meepoembedding_amd import LookupTable, Router, OPT_ADAGRAD, synth
dev = torch.device("cuda", 0)
N, dim, B, b = 100_000_000, 64, 1 << 20, 1 << 18
t = LookupTable(int(N / 0.75), dim, device=dev, max_batch=B, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, N - 8 * B, dim, dev, B)                      # leaves room for 8 batches of new keys
new = [synth.keys_t(1, N - 8 * B + s * B, B, dev) for s in range(8)]
rows = synth.rows_t(new[0], dim, 2)
old = [synth.keys_t(1, s * B, B, dev) for s in range(8)]
for k in new: t.insert(k, rows)                                       # insert, new keys
for k in old: t.insert(k, rows)                                       # insert, present keys
for k in old: t.assign(k, rows)                                       # assign
for k in old: t.find_or_insert(k)                                     # find_or_insert, all present
small = bench.lookup_batches(synth, N, b, 8, "uniform", dev, seed=3)
out = torch.empty((b, dim), dtype=torch.float32, device=dev); found = torch.empty(b, dtype=torch.uint8, device=dev)
slots = torch.empty(b, dtype=torch.int64, device=dev)
g = torch.randn(b, dim, device=dev) * 0.01
for i in range(8): t.find(small[i], out=out, found=found)             # find (headline)
for i in range(8):
    t.find_located(small[i], out=out, found=found, slots=slots)
    t.apply_adagrad(small[i], g, lr=0.01, slots=slots)                # located apply
for i in range(8): t.apply_adagrad(small[i], g, lr=0.01)              # plain apply
for i in range(8):
    t.find_located(small[i], out=out, found=found, slots=slots, prepare_apply=True)   # the training forward: find + the apply's partition in one launch
    t.apply_adagrad(small[i], g, lr=0.01, slots=slots)
for k in old: t.dedup_sum(k, rows)                                    # duplicate-key reduction with row sums (what a sharded backward runs in front of its exchange)
for k in old: t.dedup_keys(k)                                         # sync-free duplicate elimination (what the sharded lookup runs in front of its exchange)
r = Router(8, B, device=dev)
for k in old: r.partition(k)                                          # shard partition (8 owners)
for k in new[:4]: t.remove(k)                                         # remove
t.size()                                                              # size
k_, v_ = t.export(); del k_, v_                                       # export
torch.cuda.synchronize()
assert t.status() == 0

"""Quick tour of the Python host layer on one MI355X (run on the GPU box: python examples/quickstart.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__

__graft_entry__.build()
from meepoembedding_amd import INIT_UNIFORM, OPT_ADAGRAD, LookupTable, _lib  # noqa: E402
from meepoembedding_amd.nn import DynamicEmbedding  # noqa: E402
from meepoembedding_amd.tiered import TieredLookupTable  # noqa: E402

dev = torch.device("cuda", 0)

# 1. a table: int64 key -> fp32[64] row, Adagrad state, unseen keys get a hashed uniform(-0.05, 0.05) row
table = LookupTable(1 << 20, 64, device=dev, optimizer=OPT_ADAGRAD, max_batch=1 << 16, initializer=INIT_UNIFORM, init_scale=0.05)
keys = torch.randint(0, 10**12, (50_000,), device=dev)
rows, existed = table.find_or_insert(keys)               # forward lookup of a training step
table.apply_adagrad(keys, torch.randn_like(rows) * 0.01, lr=0.01)   # duplicate keys are reduced, one update per key
print("stored keys:", table.size(), "| new in this batch:", int((existed == 0).sum()))

out, found = table.find(torch.cat([keys[:5], torch.tensor([42], device=dev)]))   # inference lookup: absent -> default row
print("found mask:", found.tolist())
table.remove(keys[:1000])
ek, ev = table.export()                                    # checkpoint: int64 keys[N], fp32 values[N, 64]
print("after remove:", table.size(), "exported", tuple(ev.shape))

# 2. the same table as a torch layer (forward = find_or_insert, backward = the table's sparse Adagrad)
layer = DynamicEmbedding(table, optimizer="adagrad", lr=0.01)
ids = torch.randint(0, 10**12, (256, 8), device=dev)
loss = layer(ids).sum(1).pow(2).mean()
loss.backward()
print("layer output", tuple(layer(ids).shape), "| table now holds", table.size())

# 3. hot/cold pair: rows of the cold table live in pinned host DRAM, the kernels read them over PCIe
hot = LookupTable(1 << 16, 64, device=dev, max_batch=1 << 16, track_hits=True)
cold = LookupTable(1 << 20, 64, device=dev, max_batch=1 << 16, value_memory=_lib.MEM_HOST_PINNED, track_hits=True)
tiered = TieredLookupTable(hot, cold, hot_key_limit=40_000)
tiered.insert(keys, rows)                                   # 50K new keys > the hot room of 40K: the batch goes to the cold tier
for _ in range(8):
    tiered.find(keys[torch.randint(0, 5000, (20_000,), device=dev)])   # a hot working set of 5000 keys
print("rebalance (promoted, demoted):", tiered.rebalance(), "| hot", hot.size(), "cold", cold.size())

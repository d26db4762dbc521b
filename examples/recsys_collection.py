"""A recommender's sparse part on one MI355X: 26 tables, one pooled lookup launch and one optimizer step per batch
(run on the GPU box: python examples/recsys_collection.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__

__graft_entry__.build()
from meepoembedding_amd import INIT_UNIFORM, OPT_ADAGRAD, LookupTable, TableGroup  # noqa: E402
from meepoembedding_amd.nn import DynamicEmbeddingBag  # noqa: E402

dev = torch.device("cuda", 0)
n_tables, dim, batch, ids_per_bag, vocab = 26, 64, 2048, 5, 10**6

tables = [LookupTable(2 * vocab, dim, device=dev, optimizer=OPT_ADAGRAD, max_batch=1 << 18, initializer=INIT_UNIFORM,
                      init_scale=0.05, init_seed=j) for j in range(n_tables)]
group = TableGroup(tables, max_apply_batch=n_tables * batch * 2 * ids_per_bag)   # an upper bound on the ids of one step
sparse = DynamicEmbeddingBag(group, mode="sum", optimizer="adagrad", lr=0.05, create_missing=True).to(dev)   # bag b -> table b // batch; new ids enter their tables in forward
dense = torch.nn.Sequential(torch.nn.Linear(n_tables * dim, 256), torch.nn.ReLU(), torch.nn.Linear(256, 1)).to(dev)
dense_opt = torch.optim.SGD(dense.parameters(), lr=0.01)

for step in range(5):
    # the input pipeline's "jagged" format: all ids concatenated table by table, one offset per (table, sample) bag
    lens = torch.randint(1, 2 * ids_per_bag, (n_tables * batch,), device=dev)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)])
    ids = torch.randint(0, vocab, (int(offsets[-1]),), device=dev)
    labels = torch.rand(batch, 1, device=dev)
    pooled = sparse(ids, offsets)                                        # [n_tables * batch, dim]: 3 launches create new ids, ONE does the pooled lookup
    features = pooled.view(n_tables, batch, dim).transpose(0, 1).reshape(batch, n_tables * dim)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(dense(features), labels)
    dense_opt.zero_grad()
    loss.backward()                                                      # the sparse update (7 launches) happens in here
    dense_opt.step()
    print(f"step {step}: loss {loss.item():.4f}, ids {ids.numel()}, keys stored {sum(t.size() for t in tables)}")

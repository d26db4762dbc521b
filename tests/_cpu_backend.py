"""Test-only adapters: the CPU oracle behind the LookupTable / Router method names, on CPU torch tensors, so the
sharded host logic (meepoembedding_amd/sharded.py) can run under gloo without a GPU.  Never used by the product."""
import numpy as np
import torch

import oracle


class CpuTable:
    def __init__(self, capacity, dim, **kw):
        track = kw.pop("track_hits", False)
        self.o = oracle.OracleTable(capacity, dim, **kw)
        kw["track_hits"] = track
        self.dim = dim
        self.optimizer = kw.get("optimizer", 0)
        self.device = torch.device("cpu")
        self.capacity = self.o.capacity
        self.track_hits = bool(kw.pop("track_hits", False)) if "track_hits" in kw else False
        self.hits = {}

    def find(self, keys):
        out, found = self.o.find(keys.numpy())
        return torch.from_numpy(out), torch.from_numpy(found)

    def find_plane(self, plane, keys):
        out, found = self.o.find_plane(plane, keys.numpy())
        return torch.from_numpy(out), torch.from_numpy(found)

    def assign_plane(self, plane, keys, values):
        return torch.from_numpy(self.o.assign_plane(plane, keys.numpy(), values.numpy()))

    def remove(self, keys):
        return torch.from_numpy(self.o.remove(keys.numpy()))

    def dedup_sum(self, keys, grads=None, compact=True):   # (the oracle's form is the compact one)
        uniq, gs, inv, cnt = oracle.dedup_sum(keys.numpy(), None if grads is None else grads.numpy(), self.dim)
        return torch.from_numpy(uniq), (None if grads is None else torch.from_numpy(gs)), torch.from_numpy(cnt.astype(np.int32)), torch.from_numpy(inv)

    def find_counted(self, keys, out=None, found=None, missing_only=False):
        if missing_only:
            before = found.clone()
            self.find_missing(keys, out, found)
            hit = (before == 0) & (found != 0)
        else:
            out, found = self.find(keys)
            hit = found != 0
        for k in keys[hit].tolist():
            self.hits[k] = self.hits.get(k, 0) + 1
        return out, found

    def hits_scan(self, min_hits, max_hits, cap, reset=False):
        stored = self.o.export()[0].tolist()
        out = [k for k in stored if min_hits <= self.hits.get(k, 0) <= max_hits][:cap]
        if reset:
            self.hits = {}
        return torch.tensor(out, dtype=torch.int64)

    def insert_missing(self, keys, values, found):
        m = torch.nonzero(found == 0).view(-1)
        if m.numel():
            self.insert(keys[m], values[m])

    def find_or_insert_missing(self, keys, out, found):
        m = torch.nonzero(found == 0).view(-1)
        if m.numel():
            rows, _ = self.find_or_insert(keys[m])
            out[m] = rows

    def find_missing(self, keys, out, found):
        miss = torch.nonzero(found == 0).view(-1)
        if miss.numel():
            rows, f = self.find(keys[miss])
            hit = f != 0
            out[miss[hit]] = rows[hit]
            found[miss[hit]] = 1

    def find_or_insert(self, keys):
        out, found = self.o.find_or_insert(keys.numpy())
        return torch.from_numpy(out), torch.from_numpy(found)

    def insert(self, keys, values):
        self.o.insert(keys.numpy(), values.numpy())

    def assign(self, keys, values):
        return torch.from_numpy(self.o.assign(keys.numpy(), values.numpy()))

    def apply_adagrad(self, keys, grads, lr, eps=1e-10):
        self.o.apply_adagrad(keys.numpy(), grads.numpy(), lr, eps)

    def apply_adam(self, keys, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
        self.o.apply_adam(keys.numpy(), grads.numpy(), lr, beta1, beta2, eps, step)

    def size(self):
        return self.o.size()

    def export(self, with_state=False):
        return tuple(torch.from_numpy(x) if x is not None else None for x in self.o.export(with_state=with_state))

    def iter_export(self, chunk_slots=1 << 22, with_state=True):
        k, v, a, b = self.export(with_state=True)
        for s in range(0, k.numel(), 1000):   # several pieces, like a ranged export
            yield k[s:s + 1000], v[s:s + 1000], (a[s:s + 1000] if a is not None else None), (b[s:s + 1000] if b is not None else None)

    def import_(self, keys, values, state1=None, state2=None):
        self.insert(keys, values)
        for plane, st in ((1, state1), (2, state2)):
            if st is not None:
                self.assign_plane(plane, keys, st)


class CpuRouter:
    def __init__(self, n_shards):
        self.n_shards = n_shards

    def partition(self, keys):
        send, counts, perm = oracle.partition(keys.numpy(), self.n_shards)
        return torch.from_numpy(send), torch.from_numpy(counts), torch.from_numpy(perm)

    def owner(self, keys):
        return torch.from_numpy(oracle.hash_batch(keys.numpy(), 1, self.n_shards)[2].astype(np.int64))

    def scatter_rows(self, rows, perm):
        out = torch.empty_like(rows)
        out[perm] = rows
        return out

    def gather_rows(self, rows, perm, n_out=None):
        return rows[perm].contiguous()

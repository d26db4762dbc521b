"""Row-sharded lookup table over one node's GPUs: owner(key) routing + all-to-all exchange (SPEC.md §5).

One process per GPU; `torch.distributed` carries the exchange (backend "nccl" is RCCL over xGMI on ROCm; "gloo"
is used by the CPU tests of this host logic).  Per sharded find:

    partition(keys) by owner  ->  all-to-all counts  ->  all-to-all keys (8 B each)
    local find on the received keys (HIP kernel)     ->  all-to-all rows (dim*4 B each) + found-mask back
    scatter through perm into batch order

Reference anchor: /root/reference/README.md:2 ("A distributed … Embedding"); the snapshot has no code.

This module is host logic only: `local` is any object with the LookupTable operator methods and `router` any
object with partition / gather_rows / scatter_rows (the HIP-backed LookupTable / Router in production).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class ShardedLookupTable:
    def __init__(self, local, router, group=None):
        self.local, self.router, self.group = local, router, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # a gloo group cannot move device tensors: stage them through host memory (rehearsals of the multi-rank GPU
        # path on a single-GPU box; production uses the "nccl" backend = RCCL over xGMI, no staging)
        self._stage = dist.get_backend(group) == "gloo"
        self.dim = local.dim
        if router.n_shards != self.world:
            raise ValueError(f"router has {router.n_shards} shards, process group has {self.world} ranks")

    # -- exchange plumbing -------------------------------------------------------------------------------
    def _route(self, keys: torch.Tensor):
        """partition + counts exchange. Returns (send_keys, perm, send_splits, recv_splits)."""
        send_keys, counts, perm = self.router.partition(keys)
        if self._stage and counts.is_cuda:
            c_host = counts.cpu()
            r_host = torch.empty_like(c_host)
            dist.all_to_all_single(r_host, c_host, group=self.group)
            return send_keys, perm, c_host.tolist(), r_host.tolist()
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)
        both = torch.stack([counts, recv_counts]).cpu()  # the one host sync of the exchange
        return send_keys, perm, both[0].tolist(), both[1].tolist()

    def _a2a(self, t: torch.Tensor, in_splits, out_splits) -> torch.Tensor:
        if self._stage and t.is_cuda:
            src = t.contiguous().cpu()
            dst = torch.empty((sum(out_splits),) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.all_to_all_single(dst, src, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            return dst.to(t.device)
        out = torch.empty((sum(out_splits),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_to_all_single(out, t.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        return out

    # -- operators ---------------------------------------------------------------------------------------
    def _lookup(self, keys: torch.Tensor, insert_missing: bool, dedup: bool = False):
        keys = keys.contiguous().view(-1)
        if dedup:
            return self._lookup_dedup(keys, insert_missing)
        send_keys, perm, ss, rs = self._route(keys)
        recv_keys = self._a2a(send_keys, ss, rs)
        if insert_missing:
            rows, found = self.local.find_or_insert(recv_keys)
        else:
            rows, found = self.local.find(recv_keys)
        rows_back = self._a2a(rows, rs, ss)
        found_back = self._a2a(found, rs, ss)
        return self.router.scatter_rows(rows_back, perm), self.router.scatter_rows(found_back, perm)

    def _lookup_dedup(self, keys: torch.Tensor, insert_missing: bool):
        """Pre-exchange duplicate elimination: only the batch's DISTINCT keys cross xGMI (keys out, rows back); every
        occurrence is then served from its distinct key's row.  On skewed streams the link traffic scales with the
        number of unique keys while the metric counts lookups (SURVEY §7 hard part 1)."""
        uniq, _, _, inverse = self.local.dedup_sum(keys)       # the local table's group table does the grouping
        rows_u, found_u = self._lookup(uniq, insert_missing)
        # reserved keys have inverse -1: point them at an extra all-default "missing" row
        miss_row, _ = self.local.find(keys.new_full((1,), -(1 << 63)))
        rows_u = torch.cat([rows_u, miss_row])
        found_u = torch.cat([found_u, found_u.new_zeros(1)])
        inverse = torch.where(inverse < 0, torch.full_like(inverse, uniq.numel()), inverse)
        return self.router.gather_rows(rows_u, inverse, n_out=keys.numel()), self.router.gather_rows(found_u, inverse, n_out=keys.numel())

    def find(self, keys: torch.Tensor, dedup: bool = False):
        return self._lookup(keys, False, dedup)

    def find_or_insert(self, keys: torch.Tensor, dedup: bool = False):
        return self._lookup(keys, True, dedup)

    def remove(self, keys: torch.Tensor) -> torch.Tensor:
        keys = keys.contiguous().view(-1)
        send_keys, perm, ss, rs = self._route(keys)
        rk = self._a2a(send_keys, ss, rs)
        found = torch.cat([self.local.remove(rk[s:e]) for s, e in self._chunks(rk.numel())])
        return self.router.scatter_rows(self._a2a(found, rs, ss), perm)

    def _push(self, keys: torch.Tensor, payload: torch.Tensor):
        """Route (key, row) pairs to their owners; received pairs are ordered by source rank, then batch position."""
        keys = keys.contiguous().view(-1)
        send_keys, perm, ss, rs = self._route(keys)
        send_rows = self.router.gather_rows(payload.contiguous().view(keys.numel(), -1), perm)
        return self._a2a(send_keys, ss, rs), self._a2a(send_rows, ss, rs), perm, ss, rs

    # An owner may receive more pairs than its table's max_batch (skew, or simply world x batch).  insert / assign /
    # remove are sequentially consistent, so the received pairs are applied max_batch at a time (order preserved =
    # last-wins preserved).  apply_* is ONE update per distinct key and cannot be chunked: size the local table's
    # max_batch for the largest received batch (<= world x per-rank batch).
    def _chunks(self, n: int):
        step = int(getattr(self.local, "max_batch", n) or n) or 1
        return [(s, min(n, s + step)) for s in range(0, n, step)] or [(0, 0)]

    def insert(self, keys: torch.Tensor, values: torch.Tensor) -> None:
        rk, rv, *_ = self._push(keys, values)
        for s, e in self._chunks(rk.numel()):
            self.local.insert(rk[s:e], rv[s:e])

    def assign(self, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        rk, rv, perm, ss, rs = self._push(keys, values)
        found = torch.cat([self.local.assign(rk[s:e], rv[s:e]) for s, e in self._chunks(rk.numel())])
        return self.router.scatter_rows(self._a2a(found, rs, ss), perm)

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10) -> None:
        rk, rg, *_ = self._push(keys, grads)
        self.local.apply_adagrad(rk, rg, lr, eps)

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1) -> None:
        rk, rg, *_ = self._push(keys, grads)
        self.local.apply_adam(rk, rg, lr, beta1, beta2, eps, step)

    def size(self) -> int:
        t = torch.tensor([self.local.size()], dtype=torch.int64, device="cpu" if self._stage else self._dev())
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def save(self, path: str, chunk_slots: int = 1 << 22) -> int:
        """Checkpoint: every rank writes its shard under path/shard-<rank>-of-<world> (checkpoint.py); collective."""
        from . import checkpoint
        n = checkpoint.save_sharded(self, path, chunk_slots)
        dist.barrier(group=self.group)
        return n

    def load(self, path: str, chunk_pairs: int | None = None) -> int:
        """Load a checkpoint written by ANY world size: each rank keeps the pairs it owns now (re-sharding on load)."""
        from . import checkpoint
        n = checkpoint.load_sharded(self, path, self.router.owner, chunk_pairs)
        dist.barrier(group=self.group)
        return n

    def export_local(self, with_state: bool = False):
        return self.local.export(with_state=with_state)

    def _dev(self):
        return getattr(self.local, "device", torch.device("cpu"))

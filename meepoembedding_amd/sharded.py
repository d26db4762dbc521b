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
        uniq, _, _, inverse = self.local.dedup_sum(keys, compact=True)   # (this path synchronises for its split sizes anyway)
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

    def _aggregate(self, keys: torch.Tensor, grads: torch.Tensor):
        """Pre-exchange gradient aggregation: one (key, summed row) pair per distinct key of this rank's batch (fp64 sums rounded once, mee_dedup_sum) —
        on skewed streams the backward's bytes on xGMI scale with the distinct keys.  The owner's apply adds the ranks' partial sums up in fp64 again:
        within 1e-6 of the un-aggregated update (one extra rounding per rank and key)."""
        keys = keys.contiguous().view(-1)
        uniq, gsum, _, _ = self.local.dedup_sum(keys, grads.contiguous().view(keys.numel(), -1), compact=True)   # (this path synchronises for its split sizes anyway)
        return uniq, gsum

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10, dedup: bool = False) -> None:
        if dedup:
            keys, grads = self._aggregate(keys, grads)
        rk, rg, *_ = self._push(keys, grads)
        self.local.apply_adagrad(rk, rg, lr, eps)

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1, dedup: bool = False) -> None:
        if dedup:
            keys, grads = self._aggregate(keys, grads)
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


class RcclShardedTable:
    """The same row-sharded table with the whole exchange behind the C-ABI (`mee_sharded_*`, csrc/meepo_sharded.hip):
    partition, grouped ncclSend/ncclRecv of keys (+ rows), the local operator, ONE grouped exchange of rows + found bytes
    back and the un-permute all run inside the library on the caller's stream.  `torch.distributed` is only used once, to
    hand rank 0's ncclUniqueId to the other ranks; per step there is no Python-side collective at all.

    pad_slack = 0: exact message sizes (one host synchronisation per operator, for the split sizes).
    pad_slack >= 1: fixed-capacity EMPTY-padded segments, no host synchronisation (see include/meepo_embedding.h).
    cold: the cold table of a hot/cold pair (`local` = the hot one): BASELINE configs[4] behind the C-ABI (mee_sharded_create_ex).
    dedup: lookups exchange only the batch's distinct keys, applies one summed gradient row per distinct key (MEE_SHARDED_DEDUP)."""

    def __init__(self, local, max_batch: int, group=None, pad_slack: float = 0.0, cold=None, dedup: bool = False, hot_key_limit: int = 0):
        import ctypes as C

        from . import _lib
        from ._lib import check
        self._lib, self._check, self._C = _lib, check, C
        self.local, self.group = local, group
        self.dedup = bool(dedup)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device, self.dim, self.max_batch = local.device, local.dim, int(max_batch)
        L = _lib.lib()
        self._comm = self._h = None
        # rank 0 makes the ncclUniqueId; everybody joins the communicator (collective)
        ident = (C.c_char * 128)()
        if self.rank == 0:
            check(L.mee_comm_unique_id(ident))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        comm = C.c_void_p()
        check(L.mee_comm_create(box[0], self.world, self.rank, self.device.index, C.byref(comm)))
        self._comm = comm
        h = C.c_void_p()
        self.cold = cold
        opt = _lib.ShardedOptions(struct_size=C.sizeof(_lib.ShardedOptions), flags=_lib.SHARDED_DEDUP if dedup else 0, max_batch=self.max_batch,
                                  pad_slack=float(pad_slack), cold=cold._h if cold is not None else None, hot_key_limit=int(hot_key_limit))
        check(L.mee_sharded_create_ex(local._h, self._comm, C.byref(opt), C.byref(h)))
        self._h = h
        cap = C.c_uint64()
        check(L.mee_sharded_info(self._h, None, None, C.byref(cap)))
        self.segment_capacity = cap.value

    def close(self) -> None:
        L = self._lib.lib()
        if getattr(self, "_h", None):
            L.mee_sharded_destroy(self._h)
            self._h = None
        if getattr(self, "_comm", None):
            L.mee_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _s(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _k(self, keys: torch.Tensor) -> torch.Tensor:
        if keys.dtype != torch.int64 or keys.device != self.device:
            raise self._lib.MeepoError(self._lib.ERR_INVALID_ARG, f"keys must be int64 on {self.device}")
        return keys.contiguous().view(-1)

    def _r(self, rows: torch.Tensor, n: int) -> torch.Tensor:
        if rows.dtype != torch.float32 or rows.device != self.device or rows.numel() != n * self.dim:
            raise self._lib.MeepoError(self._lib.ERR_INVALID_ARG, f"rows must be float32 [{n},{self.dim}] on {self.device}")
        return rows.contiguous()

    def _lookup(self, fn, keys, out, found):
        k = self._k(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(fn(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), self._s()))
        return out, found

    def find(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None):
        return self._lookup(self._lib.lib().mee_sharded_find, keys, out, found)

    def find_or_insert(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None):
        return self._lookup(self._lib.lib().mee_sharded_find_or_insert, keys, out, found)

    def insert(self, keys: torch.Tensor, values: torch.Tensor) -> None:
        k = self._k(keys)
        v = self._r(values, k.numel())
        self._check(self._lib.lib().mee_sharded_insert(self._h, k.data_ptr(), v.data_ptr(), k.numel(), self._s()))

    def assign(self, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        k = self._k(keys)
        v = self._r(values, k.numel())
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        self._check(self._lib.lib().mee_sharded_assign(self._h, k.data_ptr(), v.data_ptr(), k.numel(), found.data_ptr(), self._s()))
        return found

    def remove(self, keys: torch.Tensor) -> torch.Tensor:
        k = self._k(keys)
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        self._check(self._lib.lib().mee_sharded_remove(self._h, k.data_ptr(), k.numel(), found.data_ptr(), self._s()))
        return found

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10) -> None:
        k = self._k(keys)
        g = self._r(grads, k.numel())
        self._check(self._lib.lib().mee_sharded_apply_adagrad(self._h, k.data_ptr(), g.data_ptr(), k.numel(), lr, eps, self._s()))

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1) -> None:
        k = self._k(keys)
        g = self._r(grads, k.numel())
        self._check(self._lib.lib().mee_sharded_apply_adam(self._h, k.data_ptr(), g.data_ptr(), k.numel(), lr, beta1, beta2, eps, step, self._s()))

    def size(self) -> int:
        n = self._C.c_size_t()
        self._check(self._lib.lib().mee_sharded_size(self._h, self._C.byref(n), self._s()))
        return n.value

    def status(self) -> int:
        b = self._C.c_uint32()
        self._check(self._lib.lib().mee_sharded_status(self._h, self._C.byref(b), self._s()))
        return b.value

    def clear_status(self) -> None:
        self._check(self._lib.lib().mee_sharded_clear_status(self._h, self._s()))

    def export_local(self, with_state: bool = False):
        a = self.local.export(with_state=with_state)
        if getattr(self, "cold", None) is None:
            return a
        b = self.cold.export(with_state=with_state)
        return tuple(None if x is None else torch.cat([x, y]) for x, y in zip(a, b))

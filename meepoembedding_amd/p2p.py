"""Sharded find over peer-mapped memory: the owners' gather kernels store rows straight into the requester's buffer.

SPEC.md §5 semantics, SURVEY.md §8f rank 2 design: instead of two RCCL all-to-alls per lookup (keys out, rows back) plus
an un-permute pass, every rank maps its peers' inboxes and result buffers through HIP IPC (`mee_p2p_*` in the C-ABI) and

    partition(keys)  ->  push: keys + batch positions stored into the owners' inboxes          (xGMI stores)
    barrier          ->  find: each owner probes its inbox and stores every row (and found byte) directly at
                               out[batch position] of the rank that asked                       (xGMI stores)
    barrier          ->  out[0:n], found[0:n] are complete, in batch order

No all-to-all, no staging buffers, no un-permute, no host sync; the two barriers are one-wave kernels over peer-mapped
flag words (`mee_p2p_barrier`; `device_barrier=False` uses one-element all-reduces on the caller's stream instead).

Mutators (payload=True) ride the same inboxes: push stores (key, row) pairs into the owners' key and row inboxes and pads
every segment to its fixed capacity with EMPTY keys (padding, SPEC.md §2), so after the barrier each owner hands its
WHOLE inbox — world x cap positions, ordered by source rank then batch position, the order the all-to-all path
delivers — to insert / assign / apply_* with a constant n: no counts exchange, no host sync, graph-capturable per rank.  Reference anchor: /root/reference/README.md:2 ("A distributed high-performance … Embedding").
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


class _Raw:
    """device memory owned by the library, exposed to torch through the CUDA array interface"""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


class PeerShardedFind:
    def __init__(self, local, router, max_batch: int, group=None, slack: float = 1.25, payload: bool = False,
                 device_barrier: bool = True):
        self.local, self.router, self.group = local, router, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device, self.dim, self.max_batch = local.device, local.dim, max_batch
        if router.n_shards != self.world:
            raise ValueError("router shards != process group size")
        # room for an uneven split: mean + 25 % + a constant (uniform hashing of 1M keys over 8 owners deviates by < 0.3 %)
        self.cap = int(max_batch / self.world * slack) + 4096
        self._staged = dist.get_backend(group) == "gloo"  # rehearsal on one GPU: no RCCL for the collective barrier
        # per-step barriers: a one-wave kernel over peer-mapped flag words (mee_p2p_barrier) instead of a collective
        self.device_barrier = device_barrier
        L = _lib.lib()
        self._h = None
        self._tok = torch.zeros(1, dtype=torch.int32, device="cpu" if self._staged else self.device)
        # set-up is collective: after every local step the ranks agree on success, so a rank whose HIP IPC call fails
        # makes ALL ranks raise together instead of leaving the others in a barrier
        err = None
        if payload and self.world * self.cap > local.max_batch:
            raise ValueError(f"payload mode hands the owner world*cap = {self.world * self.cap} positions per operator call; "
                             f"create the local table with max_batch >= that (it has {local.max_batch})")
        self.payload = payload
        mine = (C.c_char * (6 * 64))()
        try:
            h = C.c_void_p()
            check(L.mee_p2p_create(self.device.index, self.world, self.rank, self.cap, max_batch, self.dim, int(payload), C.byref(h)))
            self._h = h
            check(L.mee_p2p_export(self._h, mine))
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "create/export")
        gathered = [None] * self.world
        dist.all_gather_object(gathered, bytes(mine), group=group)
        try:
            check(L.mee_p2p_connect(self._h, b"".join(gathered)))
            po, pf = C.c_void_p(), C.c_void_p()
            check(L.mee_p2p_buffers(self._h, C.byref(po), C.byref(pf)))
            # + the spare last row / byte (never written by a lookup): where de-duplicated lookups park reserved keys
            self.out = torch.as_tensor(_Raw(po.value, (max_batch + 1, self.dim), "<f4"), device=self.device)
            self.found = torch.as_tensor(_Raw(pf.value, (max_batch + 1,), "|u1"), device=self.device)
            self.out[max_batch].fill_(float(getattr(local, "default_value", 0.0)))
            self.found[max_batch] = 0
            if payload:
                pk, pr, ns = C.c_void_p(), C.c_void_p(), C.c_uint64()
                check(L.mee_p2p_inbox(self._h, C.byref(pk), C.byref(pr), C.byref(ns)))
                self.inbox_keys = torch.as_tensor(_Raw(pk.value, (ns.value,), "<i8"), device=self.device)
                self.inbox_rows = torch.as_tensor(_Raw(pr.value, (ns.value, self.dim), "<f4"), device=self.device)
                self._ensure_found = torch.empty(ns.value, dtype=torch.uint8, device=self.device)
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "connect")  # also the barrier: nobody pushes before every rank has connected

    def _agree(self, err, what: str) -> None:
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self._tok.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            if self._h:
                _lib.lib().mee_p2p_destroy(self._h)
                self._h = None
            raise _lib.MeepoError(_lib.ERR_HIP, f"peer-to-peer set-up failed on some rank during {what}: {err!r}")

    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                self._barrier()  # peers may still be storing into this rank's buffers
            except Exception:
                pass
            _lib.lib().mee_p2p_destroy(self._h)
            self._h = None

    def _barrier(self) -> None:
        if self.device_barrier and self._h:
            check(_lib.lib().mee_p2p_barrier(self._h, torch.cuda.current_stream(self.device).cuda_stream))
        elif self._staged:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        else:
            dist.all_reduce(self._tok, group=self.group)  # stream-ordered: waits for this rank's kernels, releases when all arrived

    def find(self, keys: torch.Tensor, check_overflow: bool = True, dedup: bool = False):
        """Rows and found bytes of `keys` in batch order.  Without dedup: VIEWS into the peer-mapped buffers, valid until
        the next find.  dedup=True (skewed batches): only the batch's DISTINCT keys cross the links — the local table's
        group table eliminates the duplicates first (sync-free: the distinct keys are padded to the batch length with
        EMPTY, which the partition drops), every occurrence is then served from its key's row."""
        keys = keys.contiguous().view(-1)
        n = keys.numel()
        if n > self.max_batch:
            raise _lib.MeepoError(_lib.ERR_BATCH_TOO_LARGE, f"n={n} exceeds max_batch={self.max_batch}")
        L = _lib.lib()
        s = torch.cuda.current_stream(self.device).cuda_stream
        if dedup:
            uniq, inverse = self.local.dedup_keys(keys, miss_index=self.max_batch)   # reserved keys -> the spare default row
            send_keys, counts, perm = self.router.partition(uniq, skip_padding=True)
            check(L.mee_p2p_push(self._h, self.router._h, send_keys.data_ptr(), perm.data_ptr(), counts.data_ptr(), n, s))
            self._barrier()
            check(L.mee_p2p_find(self._h, self.local._h, s))
            self._barrier()
            if check_overflow:
                self.check()
            # the result buffers hold the distinct keys' rows at the keys' positions in `uniq`; expand through the index
            rows = self.router.gather_rows(self.out, inverse, n_out=n)
            found = self.router.gather_rows(self.found, inverse, n_out=n)
            return rows, found
        send_keys, counts, perm = self.router.partition(keys)
        check(L.mee_p2p_push(self._h, self.router._h, send_keys.data_ptr(), perm.data_ptr(), counts.data_ptr(), n, s))
        self._barrier()
        check(L.mee_p2p_find(self._h, self.local._h, s))
        self._barrier()
        if check_overflow:
            self.check()
        return self.out[:n], self.found[:n]

    def find_or_insert(self, keys: torch.Tensor, check_overflow: bool = True):
        """find that first creates absent keys at their owners (initial rows).  The found bytes are those of the lookup
        AFTER creation (1 for every non-reserved key); use ShardedLookupTable.find_or_insert for the pre-existence mask."""
        if not self.payload:
            raise _lib.MeepoError(_lib.ERR_INVALID_ARG, "find_or_insert needs PeerShardedFind(..., payload=True)")
        keys = keys.contiguous().view(-1)
        n = keys.numel()
        if n > self.max_batch:
            raise _lib.MeepoError(_lib.ERR_BATCH_TOO_LARGE, f"n={n} exceeds max_batch={self.max_batch}")
        L = _lib.lib()
        s = torch.cuda.current_stream(self.device).cuda_stream
        send_keys, counts, perm = self.router.partition(keys)
        check(L.mee_p2p_push_rows(self._h, self.router._h, send_keys.data_ptr(), perm.data_ptr(), counts.data_ptr(), None, n, s))
        self._barrier()
        self.local.find_or_insert(self.inbox_keys, out=self.inbox_rows, found=self._ensure_found)   # the row inbox is the scratch
        check(L.mee_p2p_find(self._h, self.local._h, s))
        self._barrier()
        if check_overflow:
            self.check()
        return self.out[:n], self.found[:n]

    # -- mutators over the payload inboxes --------------------------------------------------------------------------
    def _deliver(self, keys: torch.Tensor, rows: torch.Tensor, aggregate: bool = False) -> None:
        if not self.payload:
            raise _lib.MeepoError(_lib.ERR_INVALID_ARG, "mutators need PeerShardedFind(..., payload=True)")
        keys = keys.contiguous().view(-1)
        n = keys.numel()
        rows = rows.contiguous().view(n, -1)
        if n > self.max_batch or rows.shape[1] != self.dim or rows.dtype != torch.float32:
            raise _lib.MeepoError(_lib.ERR_INVALID_ARG, f"need n <= {self.max_batch} float32 rows of dim {self.dim}")
        s = torch.cuda.current_stream(self.device).cuda_stream
        if aggregate:
            # gradient rows only: one (key, fp64-summed row) pair per distinct key of this rank's batch goes to the owner (mee_dedup_sum, sync-free: the padded
            # unique list is partitioned with the padding dropped, the push walks the counts); the owner's apply sums the ranks' partial sums in fp64 again
            keys, rows, _, _ = self.local.dedup_sum(keys, rows)
            send_keys, counts, perm = self.router.partition(keys, skip_padding=True)
        else:
            send_keys, counts, perm = self.router.partition(keys)
        check(_lib.lib().mee_p2p_push_rows(self._h, self.router._h, send_keys.data_ptr(), perm.data_ptr(), counts.data_ptr(),
                                           rows.data_ptr(), n, s))
        self._barrier()

    def _done(self, check_overflow: bool) -> None:
        self._barrier()  # the owner's operator has consumed its inbox (stream order) before any peer's next push
        if check_overflow:
            self.check()

    def insert(self, keys: torch.Tensor, values: torch.Tensor, check_overflow: bool = True) -> None:
        self._deliver(keys, values)
        self.local.insert(self.inbox_keys, self.inbox_rows)
        self._done(check_overflow)

    def assign(self, keys: torch.Tensor, values: torch.Tensor, check_overflow: bool = True) -> None:
        """assign without the found mask (nothing travels back); use ShardedLookupTable.assign when the mask is needed"""
        self._deliver(keys, values)
        self.local.assign(self.inbox_keys, self.inbox_rows)
        self._done(check_overflow)

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10, check_overflow: bool = True, dedup: bool = False) -> None:
        """dedup=True: pre-exchange gradient aggregation (one summed row per distinct key of this rank's batch crosses the links; within 1e-6 of the plain path)."""
        self._deliver(keys, grads, aggregate=dedup)
        self.local.apply_adagrad(self.inbox_keys, self.inbox_rows, lr, eps)
        self._done(check_overflow)

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1, check_overflow: bool = True, dedup: bool = False) -> None:
        self._deliver(keys, grads, aggregate=dedup)
        self.local.apply_adam(self.inbox_keys, self.inbox_rows, lr, beta1, beta2, eps, step)
        self._done(check_overflow)

    def check(self) -> None:
        bits = C.c_uint32()
        check(_lib.lib().mee_p2p_status(self._h, C.byref(bits), torch.cuda.current_stream(self.device).cuda_stream))
        if bits.value & 2:
            raise _lib.MeepoError(_lib.ERR_HIP, "peer barrier timed out: a rank did not arrive within 5 s")
        if bits.value & 1:
            raise _lib.MeepoError(_lib.ERR_BATCH_TOO_LARGE, "peer inbox overflow: a rank sent more than slots_per_peer keys to one owner")


PeerShardedTable = PeerShardedFind  # the name for payload=True users: find + insert / assign / apply_* over peer-mapped memory

"""Sharded find over peer-mapped memory: the owners' gather kernels store rows straight into the requester's buffer.

SPEC.md §5 semantics, SURVEY.md §8f rank 2 design: instead of two RCCL all-to-alls per lookup (keys out, rows back) plus
an un-permute pass, every rank maps its peers' inboxes and result buffers through HIP IPC (`mee_p2p_*` in the C-ABI) and

    partition(keys)  ->  push: keys + batch positions stored into the owners' inboxes          (xGMI stores)
    barrier          ->  find: each owner probes its inbox and stores every row (and found byte) directly at
                               out[batch position] of the rank that asked                       (xGMI stores)
    barrier          ->  out[0:n], found[0:n] are complete, in batch order

No all-to-all, no staging buffers, no un-permute, no host sync; the two barriers are one-element all-reduces on the
caller's stream (RCCL).  Reference anchor: /root/reference/README.md:2 ("A distributed high-performance … Embedding").
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
    def __init__(self, local, router, max_batch: int, group=None, slack: float = 1.25):
        self.local, self.router, self.group = local, router, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device, self.dim, self.max_batch = local.device, local.dim, max_batch
        if router.n_shards != self.world:
            raise ValueError("router shards != process group size")
        # room for an uneven split: mean + 25 % + a constant (uniform hashing of 1M keys over 8 owners deviates by < 0.3 %)
        self.cap = int(max_batch / self.world * slack) + 4096
        self._staged = dist.get_backend(group) == "gloo"  # rehearsal on one GPU: host barriers instead of RCCL
        L = _lib.lib()
        self._h = None
        self._tok = torch.zeros(1, dtype=torch.int32, device="cpu" if self._staged else self.device)
        # set-up is collective: after every local step the ranks agree on success, so a rank whose HIP IPC call fails
        # makes ALL ranks raise together instead of leaving the others in a barrier
        err = None
        mine = (C.c_char * (5 * 64))()
        try:
            h = C.c_void_p()
            check(L.mee_p2p_create(self.device.index, self.world, self.rank, self.cap, max_batch, self.dim, C.byref(h)))
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
            self.out = torch.as_tensor(_Raw(po.value, (max_batch, self.dim), "<f4"), device=self.device)
            self.found = torch.as_tensor(_Raw(pf.value, (max_batch,), "|u1"), device=self.device)
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
        if self._staged:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        else:
            dist.all_reduce(self._tok, group=self.group)  # stream-ordered: waits for this rank's kernels, releases when all arrived

    def find(self, keys: torch.Tensor, check_overflow: bool = True):
        """Rows and found bytes of `keys` in batch order — VIEWS into the peer-mapped buffers, valid until the next find."""
        keys = keys.contiguous().view(-1)
        n = keys.numel()
        if n > self.max_batch:
            raise _lib.MeepoError(_lib.ERR_BATCH_TOO_LARGE, f"n={n} exceeds max_batch={self.max_batch}")
        L = _lib.lib()
        s = torch.cuda.current_stream(self.device).cuda_stream
        send_keys, counts, perm = self.router.partition(keys)
        check(L.mee_p2p_push(self._h, self.router._h, send_keys.data_ptr(), perm.data_ptr(), counts.data_ptr(), n, s))
        self._barrier()
        check(L.mee_p2p_find(self._h, self.local._h, s))
        self._barrier()
        if check_overflow:
            self.check()
        return self.out[:n], self.found[:n]

    def check(self) -> None:
        bits = C.c_uint32()
        check(_lib.lib().mee_p2p_status(self._h, C.byref(bits), torch.cuda.current_stream(self.device).cuda_stream))
        if bits.value & 1:
            raise _lib.MeepoError(_lib.ERR_BATCH_TOO_LARGE, "peer inbox overflow: a rank sent more than slots_per_peer keys to one owner")

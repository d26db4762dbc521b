"""Host-side lookup-table interface over the C-ABI (include/meepo_embedding.h).

Mirrors the operator set BASELINE.json's north_star names for the reference's GPU backend — find / insert /
assign / export (+ find_or_insert, size, sparse Adagrad/Adam apply).  The reference snapshot defines no
interface (only /root/reference/README.md:2, "dynamic lookuptable-style Embedding"); semantics are SPEC.md §3-§4.

torch is plumbing only: it owns the device buffers and the stream.  Every method takes/returns CUDA(HIP)
tensors on the table's device and enqueues on torch's current stream; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (INIT_CONSTANT, INIT_UNIFORM, OPT_ADAGRAD, OPT_ADAM, OPT_NONE, STATUS_RESERVED_KEY,  # noqa: F401
                   STATUS_TABLE_FULL, MeepoError, check)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class LookupTable:
    """One HBM-resident shard: int64 key -> fp32[dim] row (+ optimizer state planes)."""

    def __init__(self, capacity: int, dim: int, *, device: int | torch.device = 0, optimizer: int = OPT_NONE,
                 max_batch: int = 1 << 20, default_value: float = 0.0, initial_accumulator: float = 0.0,
                 initializer: int = INIT_CONSTANT, init_scale: float = 0.0, init_seed: int = 0, value_memory: int = 0, track_hits: bool = False,
                 admission: bool = False):
        L = _lib.lib()
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if dev.type != "cuda":
            raise MeepoError(_lib.ERR_NO_DEVICE, "LookupTable needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        cfg = _lib.Config(struct_size=C.sizeof(_lib.Config), device=self.device.index, capacity=capacity, dim=dim,
                          optimizer=optimizer, max_batch=max_batch, default_value=default_value,
                          initial_accumulator=initial_accumulator, initializer=initializer, init_scale=init_scale,
                          init_seed=init_seed, value_memory=value_memory,
                          flags=(_lib.FLAG_TRACK_HITS if track_hits else 0) | (_lib.FLAG_ADMISSION if admission else 0))
        self.track_hits = track_hits
        self.default_value = float(default_value)
        h = C.c_void_p()
        self._h = None
        check(L.mee_table_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._opts = dict(device=self.device, optimizer=optimizer, max_batch=max_batch, default_value=default_value,
                          initial_accumulator=initial_accumulator, initializer=initializer, init_scale=init_scale, init_seed=init_seed,
                          value_memory=value_memory, track_hits=track_hits, admission=admission)
        info = _lib.TableInfo()
        check(L.mee_table_info_get(self._h, C.byref(info)))
        self.capacity, self.n_buckets, self.max_batch = info.capacity, info.n_buckets, info.max_batch
        self.dim, self.optimizer = info.dim, info.optimizer
        self.table_bytes, self.workspace_bytes = info.table_bytes, info.workspace_bytes
        # bumped by every call that can move or free a stored row (remove / clear / reserve): slot handles taken before it
        # (the `located` rows a pooled forward hands to its backward) are stale afterwards
        self.layout_epoch = 0

    # -- lifetime ----------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().mee_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------------------------------------
    def _keys(self, keys: torch.Tensor) -> torch.Tensor:
        if keys.dtype != torch.int64 or keys.device != self.device:
            raise MeepoError(_lib.ERR_INVALID_ARG, f"keys must be int64 on {self.device}")
        return keys.contiguous().view(-1)

    def _rows(self, rows: torch.Tensor, n: int) -> torch.Tensor:
        if rows.dtype != torch.float32 or rows.device != self.device or rows.numel() != n * self.dim:
            raise MeepoError(_lib.ERR_INVALID_ARG, f"rows must be float32 [{n},{self.dim}] on {self.device}")
        return rows.contiguous()

    def _s(self) -> int:
        return _stream_ptr(self.device)

    # -- operators (SPEC.md §3) ----------------------------------------------------------------------------
    def find(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
             want_found: bool = True, unordered: bool = False, flags: int | None = None):
        """unordered=True (mee_find_unordered): the launch is not ordered behind EARLIER work of the current stream — only for independent
        requests whose keys are complete and whose output buffers nothing earlier in the stream still touches.
        flags (mee_find_ex): this call's cache policy, an OR of _lib.FIND_* — e.g. FIND_STREAM_STORES for result buffers that rotate."""
        if flags is not None and unordered:
            raise ValueError("find(flags=..., unordered=True): mee_find_ex and mee_find_unordered are separate entry points; pass one of the two")
        k = self._keys(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None and want_found:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        if flags is not None:
            check(_lib.lib().mee_find_ex(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr() if found is not None else None, int(flags), self._s()))
            return out, found
        fn = _lib.lib().mee_find_unordered if unordered else _lib.lib().mee_find
        check(fn(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr() if found is not None else None, self._s()))
        return out, found

    def find_many(self, requests):
        """Several lookups of this table in ONE launch (mee_find_many): `requests` = up to 16 key tensors, or (keys, out, found) tuples with
        preallocated outputs.  Returns [(out, found), …] — each exactly what find() returns for that request."""
        reqs, res = (_lib.FindRequest * len(requests))(), []
        for q, r in enumerate(requests):
            keys, out, found = (r if isinstance(r, tuple) else (r, None, None))
            k = self._keys(keys)
            n = k.numel()
            if out is None:
                out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
            if found is None:
                found = torch.empty(n, dtype=torch.uint8, device=self.device)
            reqs[q] = _lib.FindRequest(k.data_ptr(), n, out.data_ptr(), found.data_ptr())
            res.append((out, found, k))   # k: keeps a contiguous copy alive until the launch is enqueued
        check(_lib.lib().mee_find_many(self._h, reqs, len(requests), self._s()))
        return [(o, f) for o, f, _ in res]

    def find_pooled(self, keys: torch.Tensor, bag_offsets: torch.Tensor, mode: str = "sum", out: torch.Tensor | None = None,
                    found: torch.Tensor | None = None):
        """Embedding-bag lookup: bag b = keys[bag_offsets[b]:bag_offsets[b+1]] (int64 offsets on the device) -> ([n_bags, dim]
        sums or means in position order, per-key found mask).  One output row per bag is written instead of one per key."""
        k = self._keys(keys) if keys.numel() else keys
        if bag_offsets.device != self.device or bag_offsets.dtype not in (torch.int64, torch.uint64) or not bag_offsets.is_contiguous() \
                or bag_offsets.numel() < 1:
            raise MeepoError(_lib.ERR_INVALID_ARG, f"bag_offsets must be contiguous int64 on {self.device} with n_bags + 1 entries")
        n_bags = bag_offsets.numel() - 1
        if out is None:
            out = torch.empty((n_bags, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_find_pooled(self._h, k.data_ptr(), k.numel(), bag_offsets.data_ptr(), n_bags, out.data_ptr(), found.data_ptr(),
                                         {"sum": 0, "mean": 1}[mode], self._s()))
        return out, found

    def find_missing(self, keys: torch.Tensor, out: torch.Tensor, found: torch.Tensor) -> None:
        """Second-tier pass: fill the positions an earlier find (on another table) left with found == 0."""
        k = self._keys(keys)
        check(_lib.lib().mee_find_missing(self._h, k.data_ptr(), k.numel(), out.data_ptr(), found.data_ptr(), self._s()))

    def find_counted(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
                     missing_only: bool = False):
        """find (or find_missing) that also bumps the hit counter of every key it finds (track_hits tables)."""
        k = self._keys(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_find_counted(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), int(missing_only), self._s()))
        return out, found

    def hits_scan(self, min_hits: int, max_hits: int, cap: int, reset: bool = False) -> torch.Tensor:
        """Keys whose hit counter lies in [min_hits, max_hits] (at most cap); reset zeroes all counters."""
        out = torch.empty(max(cap, 1), dtype=torch.int64, device=self.device)
        n = C.c_size_t()
        check(_lib.lib().mee_hits_scan(self._h, min_hits, min(max_hits, 0xFFFFFFFF), int(reset), out.data_ptr(), cap, C.byref(n), self._s()))
        return out[: min(n.value, cap)]

    def insert(self, keys: torch.Tensor, values: torch.Tensor) -> None:
        k = self._keys(keys)
        v = self._rows(values, k.numel())
        check(_lib.lib().mee_insert(self._h, k.data_ptr(), v.data_ptr(), k.numel(), self._s()))

    def insert_missing(self, keys: torch.Tensor, values: torch.Tensor, found: torch.Tensor) -> None:
        """insert restricted to the positions with found == 0 (tiered pairs: keys found in neither table)."""
        k = self._keys(keys)
        v = self._rows(values, k.numel())
        check(_lib.lib().mee_insert_missing(self._h, k.data_ptr(), v.data_ptr(), k.numel(), found.data_ptr(), self._s()))

    def find_or_insert_missing(self, keys: torch.Tensor, out: torch.Tensor, found: torch.Tensor) -> None:
        """find_or_insert restricted to the positions with found == 0; rows of those positions are written into out."""
        k = self._keys(keys)
        check(_lib.lib().mee_find_or_insert_missing(self._h, k.data_ptr(), k.numel(), out.data_ptr(), found.data_ptr(), self._s()))

    def assign(self, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        k = self._keys(keys)
        v = self._rows(values, k.numel())
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_assign(self._h, k.data_ptr(), v.data_ptr(), k.numel(), found.data_ptr(), self._s()))
        return found

    def find_plane(self, plane: int, keys: torch.Tensor):
        """find on one plane: 0 = values, 1 = acc | m, 2 = v."""
        k = self._keys(keys)
        n = k.numel()
        out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        found = torch.empty(n, dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_find_plane(self._h, plane, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), self._s()))
        return out, found

    def assign_plane(self, plane: int, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        k = self._keys(keys)
        v = self._rows(values, k.numel())
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_assign_plane(self._h, plane, k.data_ptr(), v.data_ptr(), k.numel(), found.data_ptr(), self._s()))
        return found

    def remove(self, keys: torch.Tensor) -> torch.Tensor:
        k = self._keys(keys)
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_remove(self._h, k.data_ptr(), k.numel(), found.data_ptr(), self._s()))
        self.layout_epoch += 1
        return found

    def find_or_insert(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
                       min_count: int | None = None):
        """min_count (tables created with admission=True): an absent key is created only once the table's count-min sketch has seen it
        requested at least min_count times (this batch included); until then its positions return the default row."""
        k = self._keys(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        if min_count is not None:
            check(_lib.lib().mee_find_or_insert_admit(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), int(min_count), self._s()))
        else:
            check(_lib.lib().mee_find_or_insert(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), self._s()))
        return out, found

    def find_or_insert_located(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
                               slots: torch.Tensor | None = None, prepare_apply: bool = False):
        """find_or_insert() that also returns where every key lives now (-1: reserved key / table full): the handles for
        apply_*(…, slots=…) of the same training step — the forward of a step over a growing vocabulary.
        prepare_apply: as in find_located — the launch also partitions the batch for the apply of the SAME `keys` tensor that must follow."""
        k = self._keys(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        if slots is None:
            slots = torch.empty(n, dtype=torch.int64, device=self.device)
        fn = _lib.lib().mee_find_or_insert_located_prepare if prepare_apply else _lib.lib().mee_find_or_insert_located
        check(fn(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), slots.data_ptr(), self._s()))
        return out, found, slots

    def admission_decay(self, shift: int = 1) -> None:
        """Every counter of the admission sketch >>= shift (>= 32: reset): starts a new observation window."""
        check(_lib.lib().mee_admission_decay(self._h, int(shift), self._s()))

    def size(self) -> int:
        n = C.c_size_t()
        check(_lib.lib().mee_size(self._h, C.byref(n), self._s()))
        return n.value

    def status(self) -> int:
        b = C.c_uint32()
        check(_lib.lib().mee_status(self._h, C.byref(b), self._s()))
        return b.value

    def locate(self, keys: torch.Tensor):
        """-> (slot handles [n] int64, -1 = absent; found [n] uint8).  Probe only: no row is read."""
        k = self._keys(keys)
        slots = torch.empty(k.numel(), dtype=torch.int64, device=self.device)
        found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_locate(self._h, k.data_ptr(), k.numel(), slots.data_ptr(), found.data_ptr(), self._s()))
        return slots, found

    def plane_host_view(self, plane: int = 0):
        """numpy view [capacity, dim] float32 of one plane of a MEM_HOST_PINNED table (pinned host DRAM the device also maps): what a
        staged cold-tier transfer gathers from on the host.  Valid until reserve() / close(); the caller orders its reads."""
        import numpy as np
        ptr, stride, mem = C.c_void_p(), C.c_uint64(), C.c_uint32()
        check(_lib.lib().mee_table_plane(self._h, plane, C.byref(ptr), C.byref(stride), C.byref(mem)))
        if mem.value != _lib.MEM_HOST_PINNED:
            raise MeepoError(_lib.ERR_UNSUPPORTED, "plane_host_view: the table's planes live in HBM")
        buf = (C.c_float * (self.capacity * self.dim)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(self.capacity, self.dim)

    def probe_length(self, keys: torch.Tensor) -> float:
        """Mean number of buckets a find visits for `keys` (measurement aid); synchronises."""
        k = self._keys(keys)
        tot = C.c_uint64()
        check(_lib.lib().mee_probe_length(self._h, k.data_ptr(), k.numel(), C.byref(tot), self._s()))
        return tot.value / max(1, k.numel())

    def probe_histogram(self, keys: torch.Tensor) -> list[int]:
        """[lookups of `keys` that visit 1, 2, 3, 4-or-more buckets] (mee_probe_histogram; reserved keys visit none); synchronises."""
        k = self._keys(keys)
        h = (C.c_uint64 * 4)()
        check(_lib.lib().mee_probe_histogram(self._h, k.data_ptr(), k.numel(), h, self._s()))
        return [int(x) for x in h]

    def clear_status(self) -> None:
        check(_lib.lib().mee_clear_status(self._h, self._s()))

    def clear(self) -> None:
        check(_lib.lib().mee_clear(self._h, self._s()))
        self.layout_epoch += 1

    def set_tuning(self, name: str, value: int) -> None:
        """Performance knob (never changes results): see mee_set_tuning in the header."""
        check(_lib.lib().mee_set_tuning(self._h, name.encode(), int(value)))

    def export(self, with_state: bool = False):
        """All (key, row) pairs in unspecified order; with_state also returns the optimizer planes."""
        n = self.size()
        keys = torch.empty(n, dtype=torch.int64, device=self.device)
        vals = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        s1 = torch.empty((n, self.dim), dtype=torch.float32, device=self.device) if with_state and self.optimizer != OPT_NONE else None
        s2 = torch.empty((n, self.dim), dtype=torch.float32, device=self.device) if with_state and self.optimizer == OPT_ADAM else None
        m = C.c_size_t()
        check(_lib.lib().mee_export(self._h, keys.data_ptr(), vals.data_ptr(), s1.data_ptr() if s1 is not None else None,
                                    s2.data_ptr() if s2 is not None else None, n, C.byref(m), self._s()))
        if m.value != n:
            raise MeepoError(_lib.ERR_HIP, f"export wrote {m.value} pairs, size() said {n}")
        return (keys, vals, s1, s2) if with_state else (keys, vals)

    def export_range(self, slot_begin: int, slot_end: int, with_state: bool = False):
        """The pairs stored in slots [slot_begin, slot_end) — export in bounded pieces (checkpoints, rehash of a table
        that fills most of HBM).  Always returns (keys, values, state1 | None, state2 | None)."""
        slot_end = min(slot_end, self.capacity)
        cap = max(slot_end - slot_begin, 0)
        keys = torch.empty(cap, dtype=torch.int64, device=self.device)
        vals = torch.empty((cap, self.dim), dtype=torch.float32, device=self.device)
        s1 = torch.empty((cap, self.dim), dtype=torch.float32, device=self.device) if with_state and self.optimizer != OPT_NONE else None
        s2 = torch.empty((cap, self.dim), dtype=torch.float32, device=self.device) if with_state and self.optimizer == OPT_ADAM else None
        m = C.c_size_t()
        check(_lib.lib().mee_export_range(self._h, slot_begin, slot_end, keys.data_ptr(), vals.data_ptr(),
                                          s1.data_ptr() if s1 is not None else None, s2.data_ptr() if s2 is not None else None,
                                          cap, C.byref(m), self._s()))
        n = m.value
        return keys[:n], vals[:n], (s1[:n] if s1 is not None else None), (s2[:n] if s2 is not None else None)

    def iter_export(self, chunk_slots: int = 1 << 22, with_state: bool = True):
        """Generator over export_range pieces covering the whole table (empty pieces are skipped)."""
        for b in range(0, self.capacity, chunk_slots):
            piece = self.export_range(b, b + chunk_slots, with_state)
            if piece[0].numel():
                yield piece

    def evict(self, max_hits: int = 0, limit: int = 1 << 20, reset: bool = True) -> int:
        """Eviction by the hit counters (track_hits tables): remove up to `limit` keys looked up at most `max_hits`
        times since the counters were last reset; returns how many were removed.  Slots become reusable tombstones."""
        victims = self.hits_scan(0, max_hits, limit, reset=False)
        removed = 0
        for s in range(0, victims.numel(), self.max_batch):
            removed += int(self.remove(victims[s:s + self.max_batch]).sum().item())
        if reset:
            self.hits_scan(1, 0, 0, reset=True)   # empty range: nothing listed, counters zeroed
        return removed

    def save(self, path: str, chunk_slots: int = 1 << 22, extra: dict | None = None) -> int:
        """Write a checkpoint directory (see checkpoint.py for the format); returns the number of pairs written."""
        from . import checkpoint
        return checkpoint.save_table(self, path, chunk_slots, extra)

    def load(self, path: str, chunk_pairs: int | None = None, keep=None) -> int:
        """Bulk-load a checkpoint directory into this table (rows + optimizer planes); returns pairs loaded."""
        from . import checkpoint
        return checkpoint.load_into(self, path, chunk_pairs, keep)

    def import_(self, keys: torch.Tensor, values: torch.Tensor, state1: torch.Tensor | None = None,
                state2: torch.Tensor | None = None) -> None:
        """Inverse of export(with_state=True): bulk-load pairs (and optimizer state) of any length, max_batch at a time.
        The on-disk checkpoint format is exactly export's arrays: int64 keys[N], fp32 values[N, dim] (+ state planes)."""
        k = self._keys(keys)
        n = k.numel()
        v = self._rows(values, n)
        planes = [(1, state1), (2, state2)]
        for s in range(0, n, self.max_batch):
            e = min(n, s + self.max_batch)
            self.insert(k[s:e], v[s:e])
            for plane, st in planes:
                if st is not None:
                    self.assign_plane(plane, k[s:e], self._rows(st, n)[s:e])

    def reserve(self, capacity: int) -> None:
        """Rehash IN PLACE to at least `capacity` slots (device-to-device; old and new planes must fit together)."""
        check(_lib.lib().mee_reserve(self._h, int(capacity), self._s()))
        self.layout_epoch += 1
        info = _lib.TableInfo()
        check(_lib.lib().mee_table_info_get(self._h, C.byref(info)))
        self.capacity, self.n_buckets, self.table_bytes = info.capacity, info.n_buckets, info.table_bytes

    def maybe_grow(self, max_load: float = 0.8, factor: float = 2.0) -> bool:
        """The 'dynamic' in dynamic table, as an explicit between-steps call: when size()/capacity exceeds max_load,
        reserve(factor x capacity).  Synchronises (size)."""
        if self.size() <= max_load * self.capacity:
            return False
        self.reserve(int(self.capacity * factor))
        return True

    def resized(self, capacity: int, chunk: int = 1 << 22) -> "LookupTable":
        """Rehash into a NEW table of another capacity (same options): export -> import_, state planes included.
        The table itself never resizes (SPEC.md §2); growing is this explicit copy, which needs both tables to fit
        (plus one `chunk`-slot export piece)."""
        opts = self._opts.copy()
        new = LookupTable(capacity, self.dim, **opts)
        for ek, ev, e1, e2 in self.iter_export(chunk, with_state=True):   # bounded scratch: one slot range at a time
            new.import_(ek, ev, e1, e2)
        return new

    # -- sparse optimizers (SPEC.md §4) --------------------------------------------------------------------
    def _grad_index(self, grad_index: torch.Tensor, n: int) -> torch.Tensor:
        if grad_index.device != self.device or grad_index.numel() != n:
            raise MeepoError(_lib.ERR_INVALID_ARG, f"grad_index must hold one entry per key on {self.device}")
        return grad_index.to(torch.int32).contiguous()   # read as uint32 by the kernels: indices are < 2^31

    def _slots(self, slots: torch.Tensor, n: int) -> torch.Tensor:
        if slots.dtype != torch.int64 or slots.device != self.device or slots.numel() != n:
            raise MeepoError(_lib.ERR_INVALID_ARG, f"slots must be the int64 handles find_located returned for these keys, on {self.device}")
        return slots.contiguous()

    def find_located(self, keys: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
                     slots: torch.Tensor | None = None, prepare_apply: bool = False):
        """find() that also returns each key's slot handle (-1 = absent) for apply_*(…, slots=…) of the same training step.
        prepare_apply: the training forward (mee_find_located_prepare) — the launch also partitions the batch for the apply of the SAME
        `keys` tensor that must follow (its grad-independent half runs beside the row gather)."""
        k = self._keys(keys)
        n = k.numel()
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        if slots is None:
            slots = torch.empty(n, dtype=torch.int64, device=self.device)
        fn = _lib.lib().mee_find_located_prepare if prepare_apply else _lib.lib().mee_find_located
        check(fn(self._h, k.data_ptr(), n, out.data_ptr(), found.data_ptr(), slots.data_ptr(), self._s()))
        return out, found, slots

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10,
                      grad_index: torch.Tensor | None = None, slots: torch.Tensor | None = None) -> None:
        """grad_index (optional, one int per key): position i takes row grad_index[i] of grads (pooled lookups: the bag).
        slots (optional): the handles find_located returned for `keys` in this step's forward (skips the probe)."""
        k = self._keys(keys)
        if slots is not None:
            g = self._rows(grads, k.numel())
            check(_lib.lib().mee_apply_adagrad_located(self._h, k.data_ptr(), self._slots(slots, k.numel()).data_ptr(), g.data_ptr(), k.numel(), lr, eps, self._s()))
            return
        if grad_index is not None:
            gi = self._grad_index(grad_index, k.numel())
            g = grads.contiguous().view(-1, self.dim)
            check(_lib.lib().mee_apply_adagrad_indexed(self._h, k.data_ptr(), g.data_ptr(), g.shape[0], gi.data_ptr(), k.numel(), lr, eps, self._s()))
            return
        g = self._rows(grads, k.numel())
        check(_lib.lib().mee_apply_adagrad(self._h, k.data_ptr(), g.data_ptr(), k.numel(), lr, eps, self._s()))

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1, grad_index: torch.Tensor | None = None, slots: torch.Tensor | None = None) -> None:
        k = self._keys(keys)
        if slots is not None:
            g = self._rows(grads, k.numel())
            check(_lib.lib().mee_apply_adam_located(self._h, k.data_ptr(), self._slots(slots, k.numel()).data_ptr(), g.data_ptr(), k.numel(), lr, beta1,
                                                    beta2, eps, step, self._s()))
            return
        if grad_index is not None:
            gi = self._grad_index(grad_index, k.numel())
            g = grads.contiguous().view(-1, self.dim)
            check(_lib.lib().mee_apply_adam_indexed(self._h, k.data_ptr(), g.data_ptr(), g.shape[0], gi.data_ptr(), k.numel(), lr, beta1, beta2,
                                                    eps, step, self._s()))
            return
        g = self._rows(grads, k.numel())
        check(_lib.lib().mee_apply_adam(self._h, k.data_ptr(), g.data_ptr(), k.numel(), lr, beta1, beta2, eps, step, self._s()))

    def apply_prepare(self, keys: torch.Tensor) -> None:
        """Group + plan the next apply of `keys` ahead of time (needs no grads; may run on a side stream).  The next
        apply_adagrad / apply_adam must be given the same tensor."""
        k = self._keys(keys)
        check(_lib.lib().mee_apply_prepare(self._h, k.data_ptr(), k.numel(), self._s()))

    def apply_discard(self) -> None:
        check(_lib.lib().mee_apply_discard(self._h, self._s()))

    def dedup_keys(self, keys: torch.Tensor, miss_index: int = -1):
        """Sync-free duplicate elimination: (uniq [n] = every distinct key once, EMPTY everywhere else — also between the keys —, inverse [n] = index into uniq, or
        miss_index for reserved keys).  How many keys are distinct stays on the device."""
        k = self._keys(keys)
        n = k.numel()
        uniq = torch.empty(n, dtype=torch.int64, device=self.device)
        inverse = torch.empty(n, dtype=torch.int64, device=self.device)
        check(_lib.lib().mee_dedup_keys(self._h, k.data_ptr(), n, uniq.data_ptr(), inverse.data_ptr(), int(miss_index), self._s()))
        return uniq, inverse

    def dedup_sum(self, keys: torch.Tensor, grads: torch.Tensor | None = None, miss_index: int = -1, compact: bool = False):
        """Duplicate-key reduction alone (mee_dedup_sum, sync-free): (uniq [n], summed rows [n, dim] | None, counts [n], inverse [n]), PADDED like
        dedup_keys — every distinct key once in `uniq`, EMPTY elsewhere (also between the keys; counts 0 there, summed rows not written), inverse[i]
        = index of keys[i] in uniq (miss_index for reserved keys).  compact=True (one host synchronisation: torch.nonzero) squeezes the padding
        out: uniq / rows / counts of exactly the distinct keys, inverse re-indexed (reserved keys keep miss_index)."""
        k = self._keys(keys)
        n = k.numel()
        g = self._rows(grads, n) if grads is not None else None
        uniq = torch.empty(n, dtype=torch.int64, device=self.device)
        gs = torch.empty((n, self.dim), dtype=torch.float32, device=self.device) if g is not None else None
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        inv = torch.empty(n, dtype=torch.int64, device=self.device)
        check(_lib.lib().mee_dedup_sum(self._h, k.data_ptr(), g.data_ptr() if g is not None else None, n, uniq.data_ptr(),
                                       gs.data_ptr() if gs is not None else None, cnt.data_ptr(), inv.data_ptr(), int(miss_index), self._s()))
        if not compact:
            return uniq, gs, cnt, inv
        keep = cnt > 0
        sel = torch.nonzero(keep).view(-1)
        new_index = torch.cumsum(keep.to(torch.int64), 0) - 1           # padded index -> compact index
        valid = k > _lib.RECLAIMED_KEY
        inv_c = torch.where(valid, new_index[inv.clamp(0, max(n - 1, 0))], inv) if n else inv
        return uniq[sel], (gs[sel] if gs is not None else None), cnt[sel], inv_c


class TableGroup:
    """One find launch for the lookups of many tables (same device, same dim): mee_find_grouped.

    keys = the tables' key batches concatenated, offsets = n_tables + 1 int64/uint64 bounds ON THE DEVICE (segment j =
    keys[offsets[j]:offsets[j+1]]).  Returns (rows [n, dim], found [n]) — identical to find() per table."""

    def __init__(self, tables, max_apply_batch: int = 0):
        self.tables = list(tables)
        if not self.tables:
            raise ValueError("a group needs at least one table")
        self.device, self.dim = self.tables[0].device, self.tables[0].dim
        arr = (C.c_void_p * len(self.tables))(*[t._h for t in self.tables])
        h = C.c_void_p()
        self._h = None
        with torch.cuda.device(self.device):
            check(_lib.lib().mee_group_create(arr, len(self.tables), int(max_apply_batch), C.byref(h)))
        self._h = h

    @property
    def layout_epoch(self):
        """Changes whenever a member table removed, cleared or rehashed rows: located handles from before are stale."""
        return tuple(t.layout_epoch for t in self.tables)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().mee_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, name: str, value: int) -> None:
        """Performance knobs of the group's own apply ("apply_kernel", "apply_skew_adapt", …: mee_set_tuning); never change results."""
        check(_lib.lib().mee_group_set_tuning(self._h, name.encode(), int(value)))

    def _check_offsets(self, offsets: torch.Tensor) -> None:
        if offsets.device != self.device or offsets.dtype not in (torch.int64, torch.uint64) or offsets.numel() != len(self.tables) + 1 \
                or not offsets.is_contiguous():
            raise MeepoError(_lib.ERR_INVALID_ARG, f"offsets must be {len(self.tables) + 1} contiguous int64 values on {self.device}")

    def apply_adagrad(self, keys: torch.Tensor, offsets: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10) -> None:
        """One sparse-Adagrad step over the jagged batch == apply_adagrad per table (max_apply_batch >= keys.numel())."""
        self._check_offsets(offsets)
        k = self.tables[0]._keys(keys) if keys.numel() else keys
        g = self.tables[0]._rows(grads, k.numel())
        check(_lib.lib().mee_group_apply_adagrad(self._h, k.data_ptr(), offsets.data_ptr(), g.data_ptr(), k.numel(), lr, eps,
                                                 _stream_ptr(self.device)))

    def apply_adam(self, keys: torch.Tensor, offsets: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9,
                   beta2: float = 0.999, eps: float = 1e-8, step: int = 1) -> None:
        self._check_offsets(offsets)
        k = self.tables[0]._keys(keys) if keys.numel() else keys
        g = self.tables[0]._rows(grads, k.numel())
        check(_lib.lib().mee_group_apply_adam(self._h, k.data_ptr(), offsets.data_ptr(), g.data_ptr(), k.numel(), lr, beta1, beta2,
                                              eps, step, _stream_ptr(self.device)))

    # -- the embedding-bag collection: bags_per_table bags per member, bag b belongs to member b // bags_per_table ---------
    def _check_bags(self, bag_offsets: torch.Tensor) -> int:
        nb = bag_offsets.numel() - 1
        if bag_offsets.device != self.device or bag_offsets.dtype not in (torch.int64, torch.uint64) or not bag_offsets.is_contiguous() \
                or nb < 0 or nb % len(self.tables):
            raise MeepoError(_lib.ERR_INVALID_ARG, f"bag_offsets must be contiguous int64 on {self.device} with n_tables * bags_per_table + 1 entries")
        return nb // len(self.tables)

    def find_pooled(self, keys: torch.Tensor, bag_offsets: torch.Tensor, mode: str = "sum", out: torch.Tensor | None = None,
                    found: torch.Tensor | None = None, located: torch.Tensor | None = None):
        """find_pooled of every member in one launch -> ([n_tables * bags_per_table, dim], per-key found mask).
        located (optional int64[n] buffer) receives the located rows for apply_pooled(located=...) of the same step."""
        bpt = self._check_bags(bag_offsets)
        k = self.tables[0]._keys(keys) if keys.numel() else keys
        if out is None:
            out = torch.empty((bpt * len(self.tables), self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(k.numel(), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mee_group_find_pooled(self._h, k.data_ptr(), k.numel(), bag_offsets.data_ptr(), bpt, out.data_ptr(), found.data_ptr(),
                                               located.data_ptr() if located is not None else None,
                                               {"sum": 0, "mean": 1}[mode], _stream_ptr(self.device)))
        return out, found

    def apply_pooled(self, keys: torch.Tensor, bag_offsets: torch.Tensor, bag_grads: torch.Tensor, bag_of_position: torch.Tensor,
                     optimizer: str, lr: float, eps: float | None = None, beta1: float = 0.9, beta2: float = 0.999, step: int = 1,
                     located: torch.Tensor | None = None) -> None:
        """Backward of find_pooled over the group: one optimizer step, position i takes row bag_of_position[i] of bag_grads.
        located = the buffer the forward find_pooled of this step filled: skips the probe pass."""
        bpt = self._check_bags(bag_offsets)
        k = self.tables[0]._keys(keys) if keys.numel() else keys
        gi = self.tables[0]._grad_index(bag_of_position, k.numel())
        g = bag_grads.contiguous()
        L, s = _lib.lib(), _stream_ptr(self.device)
        loc = located.data_ptr() if located is not None else None
        if optimizer == "adagrad":
            check(L.mee_group_apply_adagrad_pooled(self._h, k.data_ptr(), bag_offsets.data_ptr(), bpt, g.data_ptr(), gi.data_ptr(), loc, k.numel(),
                                                   lr, 1e-10 if eps is None else eps, s))
        else:
            check(L.mee_group_apply_adam_pooled(self._h, k.data_ptr(), bag_offsets.data_ptr(), bpt, g.data_ptr(), gi.data_ptr(), loc, k.numel(),
                                                lr, beta1, beta2, 1e-8 if eps is None else eps, step, s))

    def find_or_insert(self, keys: torch.Tensor, offsets: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None):
        """find that first creates absent keys in their member table (initial row / state); found = present before."""
        return self.find(keys, offsets, out, found, _fn="mee_group_find_or_insert")

    def find(self, keys: torch.Tensor, offsets: torch.Tensor, out: torch.Tensor | None = None, found: torch.Tensor | None = None,
             _fn: str = "mee_find_grouped"):
        k = self.tables[0]._keys(keys) if keys.numel() else keys
        n = k.numel()
        self._check_offsets(offsets)
        if out is None:
            out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if found is None:
            found = torch.empty(n, dtype=torch.uint8, device=self.device)
        check(getattr(_lib.lib(), _fn)(self._h, k.data_ptr(), offsets.data_ptr(), n, out.data_ptr(), found.data_ptr(),
                                       _stream_ptr(self.device)))
        return out, found


def hash_batch(keys: torch.Tensor, n_buckets: int, n_shards: int):
    """SPEC.md §1 on device: (mix64, bucket, owner) as int64/int64/int32 tensors (bit patterns of the unsigned values)."""
    k = keys.contiguous().view(-1)
    n = k.numel()
    mix = torch.empty(n, dtype=torch.int64, device=k.device)
    bkt = torch.empty(n, dtype=torch.int64, device=k.device)
    own = torch.empty(n, dtype=torch.int32, device=k.device)
    with torch.cuda.device(k.device):
        check(_lib.lib().mee_hash_batch(k.data_ptr(), n, n_buckets, n_shards, mix.data_ptr(), bkt.data_ptr(), own.data_ptr(),
                                        _stream_ptr(k.device)))
    return mix, bkt, own


class Router:
    """Shard partition / un-permute kernels (SPEC.md §5) with their workspace."""

    def __init__(self, n_shards: int, max_batch: int, device: int | torch.device = 0):
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.n_shards, self.max_batch = n_shards, max_batch
        h = C.c_void_p()
        self._h = None
        check(_lib.lib().mee_router_create(self.device.index, max_batch, n_shards, C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.lib().mee_router_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def owner(self, keys: torch.Tensor) -> torch.Tensor:
        """owner(key) under this router's shard count (SPEC.md §5), int64 per key."""
        return hash_batch(keys, 1, self.n_shards)[2].to(torch.int64)

    def partition(self, keys: torch.Tensor, skip_padding: bool = False):
        """Stable partition by owner -> (send_keys, counts, perm).  skip_padding: EMPTY keys belong to no shard (only the first
        counts.sum() entries of send_keys / perm are meaningful)."""
        k = keys.contiguous().view(-1)
        n = k.numel()
        send = torch.empty_like(k)
        counts = torch.empty(self.n_shards, dtype=torch.int64, device=self.device)
        perm = torch.empty(n, dtype=torch.int64, device=self.device)
        fn = _lib.lib().mee_partition_padded if skip_padding else _lib.lib().mee_partition
        check(fn(self._h, k.data_ptr(), n, send.data_ptr(), counts.data_ptr(), perm.data_ptr(), _stream_ptr(self.device)))
        return send, counts, perm

    def scatter_rows(self, rows: torch.Tensor, perm: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """out[perm[q]] = rows[q]"""
        r = rows.contiguous()
        n = perm.numel()
        if out is None:
            out = torch.empty_like(r)
        rb = r.numel() * r.element_size() // max(n, 1)
        with torch.cuda.device(self.device):
            check(_lib.lib().mee_scatter_rows(r.data_ptr(), perm.data_ptr(), n, rb, out.data_ptr(), _stream_ptr(self.device)))
        return out

    def gather_rows(self, rows: torch.Tensor, perm: torch.Tensor, out: torch.Tensor | None = None, n_out: int | None = None) -> torch.Tensor:
        """out[q] = rows[perm[q]] for q < perm.numel(); `rows` may have a different number of rows than `perm`."""
        r = rows.contiguous()
        n = perm.numel()
        rb = r.numel() * r.element_size() // max(r.shape[0], 1) if r.dim() >= 1 and r.shape[0] else 0
        if out is None:
            out = torch.empty((n,) + tuple(r.shape[1:]), dtype=r.dtype, device=r.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mee_gather_rows(r.data_ptr(), perm.data_ptr(), n, rb, out.data_ptr(), _stream_ptr(self.device)))
        return out

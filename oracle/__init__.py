"""ctypes binding of the CPU oracle (oracle/libmeepo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg —
never by meepoembedding_amd.  PARITY UNPINNED (reference snapshot has no implementation; see
oracle/meepo_oracle.h).  The class mirrors the product's host interface (meepoembedding_amd.table.LookupTable)
on numpy arrays so parity tests can drive both with the same calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmeepo_oracle.so")

OPT_NONE, OPT_ADAGRAD, OPT_ADAM = 0, 1, 2
INIT_CONSTANT, INIT_UNIFORM = 0, 1
STATUS_TABLE_FULL, STATUS_RESERVED_KEY = 1, 2
EMPTY_KEY = -(1 << 63)
RECLAIMED_KEY = EMPTY_KEY + 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "meepo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmeepo_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u64, u32, f32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_size_t
        L.meo_mix64.restype = u64; L.meo_mix64.argtypes = [u64]
        L.meo_mix64b.restype = u64; L.meo_mix64b.argtypes = [u64]
        L.meo_mulhi64.restype = u64; L.meo_mulhi64.argtypes = [u64, u64]
        L.meo_bucket.restype = u64; L.meo_bucket.argtypes = [C.c_int64, u64]
        L.meo_next_prime.restype = u64; L.meo_next_prime.argtypes = [u64]
        L.meo_step.restype = u64; L.meo_step.argtypes = [C.c_int64, u64]
        L.meo_owner.restype = u32; L.meo_owner.argtypes = [C.c_int64, u32]
        L.meo_hash_batch.restype = None; L.meo_hash_batch.argtypes = [vp, sz, u64, u32, vp, vp, vp]
        L.meo_create.restype = vp; L.meo_create.argtypes = [u64, u32, u32, f32, f32, u32, f32, u64]
        L.meo_destroy.restype = None; L.meo_destroy.argtypes = [vp]
        L.meo_capacity.restype = u64; L.meo_capacity.argtypes = [vp]
        L.meo_size.restype = u64; L.meo_size.argtypes = [vp]
        L.meo_status.restype = u32; L.meo_status.argtypes = [vp]
        L.meo_clear_status.restype = None; L.meo_clear_status.argtypes = [vp]
        L.meo_clear.restype = None; L.meo_clear.argtypes = [vp]
        L.meo_initial_row.restype = None; L.meo_initial_row.argtypes = [vp, C.c_int64, vp]
        L.meo_find.restype = None; L.meo_find.argtypes = [vp, vp, sz, vp, vp]
        L.meo_find_mt.restype = None; L.meo_find_mt.argtypes = [vp, vp, sz, vp, vp, C.c_int]
        L.meo_populate_synth_mt.restype = u64; L.meo_populate_synth_mt.argtypes = [vp, u64, u64, u64, u64, C.c_int]
        L.meo_insert.restype = None; L.meo_insert.argtypes = [vp, vp, vp, sz]
        L.meo_assign.restype = None; L.meo_assign.argtypes = [vp, vp, vp, sz, vp]
        L.meo_find_plane.restype = None; L.meo_find_plane.argtypes = [vp, u32, vp, sz, vp, vp]
        L.meo_assign_plane.restype = None; L.meo_assign_plane.argtypes = [vp, u32, vp, vp, sz, vp]
        L.meo_remove.restype = None; L.meo_remove.argtypes = [vp, vp, sz, vp]
        L.meo_find_or_insert.restype = None; L.meo_find_or_insert.argtypes = [vp, vp, sz, vp, vp]
        L.meo_export.restype = u64; L.meo_export.argtypes = [vp, vp, vp, vp, vp, u64]
        L.meo_apply_adagrad.restype = None; L.meo_apply_adagrad.argtypes = [vp, vp, vp, sz, f32, f32]
        L.meo_apply_adam.restype = None; L.meo_apply_adam.argtypes = [vp, vp, vp, sz, f32, f32, f32, f32, u64]
        L.meo_dedup_sum.restype = u64; L.meo_dedup_sum.argtypes = [vp, vp, sz, u32, vp, vp, vp, vp]
        L.meo_partition.restype = None; L.meo_partition.argtypes = [vp, sz, u32, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _keys(k) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(k, dtype=np.int64))


def _rows(v, n, dim) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32)).reshape(n, dim)
    return a


def hash_batch(keys, n_buckets: int, n_shards: int):
    k = _keys(keys)
    mix = np.empty(k.size, np.uint64); bkt = np.empty(k.size, np.uint64); own = np.empty(k.size, np.uint32)
    lib().meo_hash_batch(_p(k), k.size, n_buckets, n_shards, _p(mix), _p(bkt), _p(own))
    return mix, bkt, own


def partition(keys, n_shards: int):
    k = _keys(keys)
    send = np.empty_like(k); counts = np.zeros(n_shards, np.uint64); perm = np.empty(k.size, np.int64)
    lib().meo_partition(_p(k), k.size, n_shards, _p(send), _p(counts), _p(perm))
    return send, counts.astype(np.int64), perm


def dedup_sum(keys, grads, dim: int):
    k = _keys(keys)
    g = None if grads is None else _rows(grads, k.size, dim)
    uniq = np.empty(k.size, np.int64); gs = np.empty((k.size, dim), np.float32)
    inv = np.empty(k.size, np.int64); cnt = np.empty(k.size, np.uint32)
    U = lib().meo_dedup_sum(_p(k), _p(g), k.size, dim, _p(uniq), _p(gs), _p(inv), _p(cnt))
    return uniq[:U].copy(), gs[:U].copy(), inv, cnt[:U].copy()


class OracleTable:
    """Numpy-facing mirror of the operator API, backed by the C oracle."""

    def __init__(self, capacity: int, dim: int, optimizer: int = OPT_NONE, default_value: float = 0.0,
                 initial_accumulator: float = 0.0, initializer: int = INIT_CONSTANT, init_scale: float = 0.0,
                 init_seed: int = 0):
        self._h = lib().meo_create(capacity, dim, optimizer, default_value, initial_accumulator, initializer,
                                   init_scale, init_seed)
        if not self._h:
            raise ValueError("meo_create: invalid arguments or out of memory")
        self.dim = dim
        self.optimizer = optimizer

    def close(self):
        if self._h:
            lib().meo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def capacity(self) -> int:
        return lib().meo_capacity(self._h)

    def size(self) -> int:
        return lib().meo_size(self._h)

    def status(self) -> int:
        return lib().meo_status(self._h)

    def clear_status(self):
        lib().meo_clear_status(self._h)

    def clear(self):
        lib().meo_clear(self._h)

    def initial_row(self, key: int) -> np.ndarray:
        r = np.empty(self.dim, np.float32)
        lib().meo_initial_row(self._h, key, _p(r))
        return r

    def find(self, keys, threads: int = 1, out=None, found=None):
        """threads > 1: the persistent worker pool of meo_find_mt.  out / found: caller-owned result buffers to reuse (a timing loop
        that allocated 64 MB per call would measure page faults)."""
        k = _keys(keys)
        if out is None:
            out = np.empty((k.size, self.dim), np.float32)
        if found is None:
            found = np.empty(k.size, np.uint8)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == k.size * self.dim and found.size == k.size
        if threads > 1:
            lib().meo_find_mt(self._h, _p(k), k.size, _p(out), _p(found), threads)
        else:
            lib().meo_find(self._h, _p(k), k.size, _p(out), _p(found))
        return out, found

    def insert(self, keys, values):
        k = _keys(keys); v = _rows(values, k.size, self.dim)
        lib().meo_insert(self._h, _p(k), _p(v), k.size)

    def populate_synth(self, key_seed: int, start: int, count: int, row_seed: int, threads: int = 1) -> int:
        """Bulk load synth.keys_np(key_seed, start, count) with synth.rows_np(keys, dim, row_seed), generated inside the library by
        `threads` threads (the timed CPU baseline builds its 100M-key table this way); returns the number of keys placed."""
        return int(lib().meo_populate_synth_mt(self._h, key_seed & ((1 << 64) - 1), start, count, row_seed & ((1 << 64) - 1), threads))

    def assign(self, keys, values):
        k = _keys(keys); v = _rows(values, k.size, self.dim)
        found = np.empty(k.size, np.uint8)
        lib().meo_assign(self._h, _p(k), _p(v), k.size, _p(found))
        return found

    def find_plane(self, plane, keys):
        k = _keys(keys)
        out = np.empty((k.size, self.dim), np.float32); found = np.empty(k.size, np.uint8)
        lib().meo_find_plane(self._h, plane, _p(k), k.size, _p(out), _p(found))
        return out, found

    def assign_plane(self, plane, keys, values):
        k = _keys(keys); v = _rows(values, k.size, self.dim)
        found = np.empty(k.size, np.uint8)
        lib().meo_assign_plane(self._h, plane, _p(k), _p(v), k.size, _p(found))
        return found

    def remove(self, keys):
        k = _keys(keys)
        found = np.empty(k.size, np.uint8)
        lib().meo_remove(self._h, _p(k), k.size, _p(found))
        return found

    def find_or_insert(self, keys):
        k = _keys(keys)
        out = np.empty((k.size, self.dim), np.float32); found = np.empty(k.size, np.uint8)
        lib().meo_find_or_insert(self._h, _p(k), k.size, _p(out), _p(found))
        return out, found

    def export(self, with_state: bool = False):
        n = self.size()
        keys = np.empty(n, np.int64); vals = np.empty((n, self.dim), np.float32)
        s1 = np.empty((n, self.dim), np.float32) if with_state and self.optimizer != OPT_NONE else None
        s2 = np.empty((n, self.dim), np.float32) if with_state and self.optimizer == OPT_ADAM else None
        m = lib().meo_export(self._h, _p(keys), _p(vals), _p(s1), _p(s2), n)
        assert m == n
        return (keys, vals, s1, s2) if with_state else (keys, vals)

    def apply_adagrad(self, keys, grads, lr: float, eps: float):
        k = _keys(keys); g = _rows(grads, k.size, self.dim)
        lib().meo_apply_adagrad(self._h, _p(k), _p(g), k.size, lr, eps)

    def apply_adam(self, keys, grads, lr: float, beta1: float, beta2: float, eps: float, step: int):
        k = _keys(keys); g = _rows(grads, k.size, self.dim)
        lib().meo_apply_adam(self._h, _p(k), _p(g), k.size, lr, beta1, beta2, eps, step)


def pool_rows(rows: np.ndarray, bag_offsets: np.ndarray, mode: str = "sum") -> np.ndarray:
    """SPEC.md §3 find_pooled, given find's rows: per bag, fp32 additions in position order (first row copied, the rest
    added one by one), 'mean' divides by the bag length in fp32; an empty bag is zeros."""
    off = np.asarray(bag_offsets, dtype=np.int64)
    n_bags = off.size - 1
    lens = off[1:] - off[:-1]
    out = np.zeros((n_bags, rows.shape[1]), dtype=np.float32)
    for l in range(int(lens.max()) if n_bags else 0):
        m = lens > l
        r = rows[off[:-1][m] + l]
        out[m] = r if l == 0 else (out[m] + r).astype(np.float32)
    if mode == "mean":
        nz = lens > 0
        out[nz] = (out[nz] / lens[nz, None].astype(np.float32)).astype(np.float32)
    return out

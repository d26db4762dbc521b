"""Synthetic workload generators: splitmix64 key streams and key-derived rows.

Rows are a pure function of (key, column, seed) so any row of a 100M-key table can be re-derived without a
stored copy.  Two backends with identical bits: torch (runs on the GPU for the bench) and numpy (tests and the
CPU baseline).  Generators only — no table logic lives here.
"""
from __future__ import annotations

import numpy as np
import torch

_GOLDEN = 0x9E3779B97F4A7C15
_M1, _M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _s64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


# ---- torch (int64 two's complement arithmetic == uint64 wrap-around) ------------------------------------
def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64_t(x: torch.Tensor) -> torch.Tensor:
    x = x ^ _lsr(x, 30)
    x = x * _s64(_M1)
    x = x ^ _lsr(x, 27)
    x = x * _s64(_M2)
    return x ^ _lsr(x, 31)


def keys_t(seed: int, start: int, count: int, device) -> torch.Tensor:
    """keys[i] = mix64(seed + (start+i+1)*GOLDEN) as int64 — the i-th outputs of splitmix64(seed)."""
    i = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    return mix64_t(i * _s64(_GOLDEN) + _s64(seed))


def rows_t(keys: torch.Tensor, dim: int, seed: int) -> torch.Tensor:
    """rows[k, j] = float(mix64(key ^ mix64(seed+j)) >> 40) * 2^-24 - 0.5  (exact in fp32)."""
    cj = mix64_t(torch.arange(dim, dtype=torch.int64, device=keys.device) + _s64(seed))
    h = mix64_t(keys.view(-1, 1) ^ cj.view(1, -1))
    return _lsr(h, 40).to(torch.float32) * (2.0 ** -24) - 0.5


# ---- numpy (uint64) -------------------------------------------------------------------------------------
def mix64_np(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(_M1)
        x ^= x >> np.uint64(27)
        x *= np.uint64(_M2)
        x ^= x >> np.uint64(31)
    return x


def keys_np(seed: int, start: int, count: int) -> np.ndarray:
    i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = i * np.uint64(_GOLDEN) + np.uint64(seed & ((1 << 64) - 1))
    return mix64_np(z).view(np.int64)


def rows_np(keys: np.ndarray, dim: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        cj = mix64_np(np.arange(dim, dtype=np.uint64) + np.uint64(seed))
    h = mix64_np(keys.view(np.uint64).reshape(-1, 1) ^ cj.reshape(1, -1))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24) - np.float32(0.5)

"""Hot/cold tiering of one shard: an HBM table for hot keys in front of a table whose rows live in pinned host DRAM.

BASELINE.json configs[4] ("hot/cold tier: HBM + pinned host-DRAM spill"); reference anchor /root/reference/README.md:2
("Supports GPU, CPU … backends" — a GPU backend backed by a CPU-memory one).  The snapshot has no code for it.

Design (host logic only; every row still moves through the HIP kernels):
  * `hot`  = LookupTable(value_memory=MEM_HBM)          keys + rows in HBM
  * `cold` = LookupTable(value_memory=MEM_HOST_PINNED)  keys (the index) in HBM, rows/state in pinned, device-mapped host
             DRAM — the same find/insert/apply kernels read and write them over PCIe (zero-copy), asynchronously on the
             caller's stream, so a cold access costs PCIe bandwidth but no host thread and no staging copy.
  * a key lives in exactly ONE tier, so the pair behaves like one table (tests compare it with a single oracle table).
  * new keys go to the hot tier while it has room (`hot_key_limit`), else to the cold tier; `promote` / `demote` move
    keys (values AND optimizer state) between tiers.
  * placement policy (tables created with track_hits=True): every cold hit and every `sample_every`-th hot lookup bumps
    a per-slot counter inside the find kernel; `rebalance()` promotes the cold keys that were hit at least
    `promote_threshold` times in the window and, when the hot tier is full, first demotes hot keys that no sampled
    lookup touched.  It runs BETWEEN steps, on the stream the steps run on: its moves are mutators of both tables (they share the tables'
    per-batch scratch with the step's operators and read rows the next update writes), so they are ordered with them, not beside them.
  * in a training loop (`rebalance_every` = K): the pair runs `rebalance()` itself behind every K-th optimizer step — the
    forward lookups of those K steps are the observation window.

`hot` / `cold` are any objects with the LookupTable methods, so the logic also runs on the CPU test adapters.
"""
from __future__ import annotations

import torch


class TieredLookupTable:
    def __init__(self, hot, cold, hot_key_limit: int | None = None, sample_every: int = 8, promote_threshold: int = 2,
                 rebalance_every: int = 0, rebalance_max_moves: int = 1 << 20):
        if hot.dim != cold.dim:
            raise ValueError("hot and cold tables must have the same dim")
        self.hot, self.cold, self.dim = hot, cold, hot.dim
        self.optimizer = getattr(hot, "optimizer", 0)
        # keep the hot table at a load it probes fast at (SPEC.md §2: probe length grows with load)
        self.hot_key_limit = int(hot_key_limit if hot_key_limit is not None else getattr(hot, "capacity", 0) * 0.75)
        self._hot_keys_ub = 0  # upper bound on hot.size(), maintained without synchronising
        self.policy = bool(getattr(hot, "track_hits", False) and getattr(cold, "track_hits", False))
        self.sample_every, self.promote_threshold = max(1, sample_every), promote_threshold
        self._calls = 0
        # the training loop's policy knob: a rebalance behind every rebalance_every-th apply_* (0 = the caller runs rebalance() itself)
        self.rebalance_every, self.rebalance_max_moves = max(0, int(rebalance_every)), int(rebalance_max_moves)
        self._train_steps = 0
        self.rebalance_log: list[tuple[int, int, int]] = []   # (optimizer step, promoted, demoted)

    # -- helpers -----------------------------------------------------------------------------------------
    @staticmethod
    def _idx(mask: torch.Tensor) -> torch.Tensor:
        return torch.nonzero(mask, as_tuple=False).view(-1)

    def _room_for(self, n_new: int) -> bool:
        if self._hot_keys_ub + n_new > self.hot_key_limit:
            self._hot_keys_ub = self.hot.size()  # refresh the bound (synchronises; only when the bound is hit)
        return self._hot_keys_ub + n_new <= self.hot_key_limit

    # -- operators (SPEC.md §3 on the union of both tiers) ---------------------------------------------------
    def find(self, keys: torch.Tensor):
        keys = keys.contiguous().view(-1)
        if not self.policy:
            out, found = self.hot.find(keys)
            # second pass on the same buffers: the cold table fills what the hot one missed — no host sync, no compaction
            self.cold.find_missing(keys, out, found)
            return out, found
        self._calls += 1
        if self._calls % self.sample_every == 0:
            out, found = self.hot.find_counted(keys)          # sampled: hot keys that are still in use get marked
        else:
            out, found = self.hot.find(keys)
        self.cold.find_counted(keys, out, found, missing_only=True)  # cold hits are always counted (they are PCIe-bound anyway)
        return out, found

    def rebalance(self, max_moves: int = 1 << 20) -> tuple[int, int]:
        """One policy step: promote cold keys hit >= promote_threshold times since the last call, demoting untouched hot
        keys first if the hot tier is full.  Returns (promoted, demoted).  Starts a new observation window."""
        if not self.policy:
            raise RuntimeError("rebalance() needs both tiers created with track_hits=True")
        cand = self.cold.hits_scan(self.promote_threshold, 0xFFFFFFFF, max_moves, reset=True)
        self._hot_keys_ub = self.hot.size()
        room = max(0, self.hot_key_limit - self._hot_keys_ub)
        need = max(0, int(cand.numel()) - room)
        demoted = 0
        victims = self.hot.hits_scan(0, 0, need, reset=True)   # also restarts the hot tier's window
        if need and victims.numel():
            demoted = self._move(self.hot, self.cold, victims)
            self._hot_keys_ub -= demoted
            room += demoted
        promoted = self._move(self.cold, self.hot, cand[:room]) if room and cand.numel() else 0
        self._hot_keys_ub += promoted
        return promoted, demoted

    # Mutators exploit "a key lives in exactly one tier": overwrite / delete passes simply go to BOTH tables (the one
    # that does not hold the key ignores it) and the found masks are OR-ed; only the creation of new keys needs the
    # combined mask, through the *_missing operators.  Nothing here synchronises except the rare refresh of the
    # hot-tier fill bound.
    def insert(self, keys: torch.Tensor, values: torch.Tensor) -> None:
        keys = keys.contiguous().view(-1)
        values = values.contiguous().view(keys.numel(), self.dim)
        found = self.hot.assign(keys, values) | self.cold.assign(keys, values)   # present somewhere: overwritten (last wins)
        n = keys.numel()
        if self._room_for(n):                                                   # n bounds the number of new keys
            self.hot.insert_missing(keys, values, found)
            self._hot_keys_ub += n
        else:
            self.cold.insert_missing(keys, values, found)

    def assign(self, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        keys = keys.contiguous().view(-1)
        values = values.contiguous().view(keys.numel(), self.dim)
        return self.hot.assign(keys, values) | self.cold.assign(keys, values)

    def remove(self, keys: torch.Tensor) -> torch.Tensor:
        keys = keys.contiguous().view(-1)
        return self.hot.remove(keys) | self.cold.remove(keys)

    def find_or_insert(self, keys: torch.Tensor):
        keys = keys.contiguous().view(-1)
        out, found = self.find(keys)
        n = keys.numel()
        if self._room_for(n):
            self.hot.find_or_insert_missing(keys, out, found)   # inserts the hashed initial row once per distinct new key
            self._hot_keys_ub += n
        else:
            self.cold.find_or_insert_missing(keys, out, found)
        return out, found

    def _after_optimizer_step(self) -> None:
        if not (self.policy and self.rebalance_every):
            return
        self._train_steps += 1
        if self._train_steps % self.rebalance_every == 0:
            p, d = self.rebalance(self.rebalance_max_moves)
            self.rebalance_log.append((self._train_steps, int(p), int(d)))

    def apply_adagrad(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, eps: float = 1e-10) -> None:
        # a key lives in one tier and each table ignores keys it does not hold: both see the whole batch
        self.hot.apply_adagrad(keys, grads, lr, eps)
        self.cold.apply_adagrad(keys, grads, lr, eps)
        self._after_optimizer_step()

    def apply_adam(self, keys: torch.Tensor, grads: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, step: int = 1) -> None:
        self.hot.apply_adam(keys, grads, lr, beta1, beta2, eps, step)
        self.cold.apply_adam(keys, grads, lr, beta1, beta2, eps, step)
        self._after_optimizer_step()

    # duplicate reductions touch no table row, only a table's per-batch scratch: the hot table lends its own
    def dedup_keys(self, keys: torch.Tensor, miss_index: int = -1):
        return self.hot.dedup_keys(keys, miss_index=miss_index)

    def dedup_sum(self, keys: torch.Tensor, grads: torch.Tensor | None = None, **kw):
        return self.hot.dedup_sum(keys, grads, **kw)

    def size(self) -> int:
        return self.hot.size() + self.cold.size()

    def export(self, with_state: bool = False):
        a, b = self.hot.export(with_state=with_state), self.cold.export(with_state=with_state)
        return tuple(None if x is None else torch.cat([x, y]) for x, y in zip(a, b))

    # -- checkpoint (checkpoint.py works on anything with iter_export / import_) ----------------------------------
    @property
    def device(self) -> torch.device:
        return getattr(self.hot, "device", torch.device("cpu"))

    @property
    def max_batch(self) -> int:
        return min(int(getattr(self.hot, "max_batch", 1 << 20)), int(getattr(self.cold, "max_batch", 1 << 20)))

    def iter_export(self, chunk_slots: int = 1 << 22, with_state: bool = True):
        yield from self.hot.iter_export(chunk_slots, with_state)
        yield from self.cold.iter_export(chunk_slots, with_state)

    def assign_plane(self, plane: int, keys: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
        return self.hot.assign_plane(plane, keys, values) | self.cold.assign_plane(plane, keys, values)

    def import_(self, keys: torch.Tensor, values: torch.Tensor, state1: torch.Tensor | None = None,
                state2: torch.Tensor | None = None) -> None:
        """Bulk-load pairs + optimizer planes (hot while there is room, then cold), max_batch at a time."""
        keys = keys.contiguous().view(-1)
        step = self.max_batch
        for s in range(0, keys.numel(), step):
            k = keys[s:s + step]
            self.insert(k, values[s:s + step])
            for plane, st in ((1, state1), (2, state2)):
                if st is not None:
                    self.assign_plane(plane, k, st[s:s + step])

    def save(self, path: str, chunk_slots: int = 1 << 22) -> int:
        from . import checkpoint
        return checkpoint.save_table(self, path, chunk_slots)

    def load(self, path: str, chunk_pairs: int | None = None, keep=None) -> int:
        from . import checkpoint
        return checkpoint.load_into(self, path, chunk_pairs, keep)

    # -- tier migration ------------------------------------------------------------------------------------
    def _move(self, src, dst, keys: torch.Tensor) -> int:
        """Move the keys that `src` holds into `dst`, rows and optimizer state; returns how many were moved."""
        keys = torch.unique(keys.contiguous().view(-1))
        vals, found = src.find(keys)
        idx = self._idx(found != 0)
        if not idx.numel():
            return 0
        k = keys[idx]
        dst.insert(k, vals[idx])                         # state planes of new keys start at their initial values …
        n_planes = {0: 0, 1: 1, 2: 2}[self.optimizer]
        for plane in range(1, n_planes + 1):             # … and are then overwritten with the migrated state
            st, _ = src.find_plane(plane, k)
            dst.assign_plane(plane, k, st)
        src.remove(k)
        return int(k.numel())

    def _move_staged(self, keys: torch.Tensor) -> int:
        """cold -> hot through a staged transfer (BASELINE configs[4] "async hipMemcpy side stream"): probe the cold index for the slots,
        gather the rows (and state planes) out of the pinned host planes ON THE HOST into one contiguous pinned staging buffer, ship it with
        ONE asynchronous copy on a side stream, insert into the hot table.  The alternative `_move` lets the find kernel read the same
        rows straight over PCIe (zero-copy); tools/tier_bench.py measures the two against each other."""
        import numpy as np
        keys = torch.unique(keys.contiguous().view(-1))
        slots, found = self.cold.locate(keys)
        idx = self._idx(found != 0)
        if not idx.numel():
            return 0
        k = keys[idx]
        hs = slots[idx].cpu().numpy()                     # synchronises: the host needs the slot list
        n_planes = {0: 0, 1: 1, 2: 2}[self.optimizer]
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(self.device)
        staged = []
        for plane in range(0, n_planes + 1):
            host = torch.empty((hs.size, self.dim), dtype=torch.float32, pin_memory=True)
            np.take(self.cold.plane_host_view(plane), hs, axis=0, out=host.numpy())   # host-side gather of scattered 256-B rows
            with torch.cuda.stream(self._side):
                staged.append(host.to(self.device, non_blocking=True))                # one hipMemcpyAsync per plane on the side stream
        torch.cuda.current_stream(self.device).wait_stream(self._side)
        self.hot.insert(k, staged[0])
        for plane in range(1, n_planes + 1):
            self.hot.assign_plane(plane, k, staged[plane])
        self.cold.remove(k)
        return int(k.numel())

    def promote(self, keys: torch.Tensor, staged: bool = False) -> int:
        """cold -> hot for the given keys (those that are cold), as many as the hot tier has room for.  staged=True: host-side gather +
        one asynchronous copy on a side stream instead of zero-copy reads by the find kernel (see _move_staged)."""
        keys = torch.unique(keys.contiguous().view(-1))
        _, in_cold = self.cold.find(keys)
        cand = keys[self._idx(in_cold != 0)]
        if not cand.numel():
            return 0
        self._hot_keys_ub = self.hot.size()
        room = max(0, self.hot_key_limit - self._hot_keys_ub)
        cand = cand[:room]
        moved = (self._move_staged(cand) if staged else self._move(self.cold, self.hot, cand)) if cand.numel() else 0
        self._hot_keys_ub += moved
        return moved

    def demote(self, keys: torch.Tensor) -> int:
        """hot -> cold for the given keys (those that are hot)."""
        moved = self._move(self.hot, self.cold, keys)
        self._hot_keys_ub = max(0, self._hot_keys_ub - moved)
        return moved

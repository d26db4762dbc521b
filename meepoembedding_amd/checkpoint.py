"""Checkpoint format of a table: export's arrays, written as they come.

A checkpoint is a DIRECTORY:

    meta.json     {"format": "meepo-table-v1", "dim", "optimizer", "n", "planes": ["values", "state1", ...], "extra": {...}}
    keys.i64      n little-endian int64 keys, in export order (unspecified, stable within the checkpoint)
    values.f32    n x dim little-endian fp32 rows, same order
    state1.f32    n x dim Adagrad accumulator / Adam m      (tables with an optimizer)
    state2.f32    n x dim Adam v                             (Adam tables)

The table is walked in slot ranges (`mee_export_range`), so saving needs scratch for one range, not a second copy of the
table, and the pieces go to disk in order; loading reads the files back `chunk_pairs` at a time and re-inserts them
(`insert` + `assign_plane`), so a checkpoint loads into a table of ANY capacity, and — with a `keep` filter — into any
sharding: a rank of a G'-way job keeps the pairs whose owner(key, G') is itself, whatever G wrote them.

Reference anchor: /root/reference/README.md:2 ("dynamic lookuptable-style"); the snapshot defines no on-disk format, so this
one is simply the arrays `mee_export` returns.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

FORMAT = "meepo-table-v1"
_PLANES = ("values", "state1", "state2")


def _n_planes(optimizer: int) -> int:
    return 1 + (optimizer != 0) + (optimizer == 2)


def save_table(table, path: str, chunk_slots: int = 1 << 22, extra: dict | None = None) -> int:
    os.makedirs(path, exist_ok=True)
    planes = _PLANES[: _n_planes(table.optimizer)]
    n = 0
    files = [open(os.path.join(path, "keys.i64"), "wb")] + [open(os.path.join(path, p + ".f32"), "wb") for p in planes]
    try:
        for piece in table.iter_export(chunk_slots, with_state=True):
            arrays = [piece[0]] + [x for x in piece[1:] if x is not None]
            for f, a in zip(files, arrays):
                a.cpu().numpy().tofile(f)
            n += piece[0].numel()
    finally:
        for f in files:
            f.close()
    meta = {"format": FORMAT, "dim": table.dim, "optimizer": int(table.optimizer), "n": n, "planes": list(planes), "extra": extra or {}}
    tmp = os.path.join(path, "meta.json.tmp")
    with open(tmp, "w") as f:
        json.dump(meta, f)
    os.replace(tmp, os.path.join(path, "meta.json"))   # meta.json appears last: a directory without it is an unfinished save
    return n


def read_meta(path: str) -> dict:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT:
        raise ValueError(f"{path}: not a {FORMAT} checkpoint")
    return meta


def iter_pairs(path: str, chunk_pairs: int):
    """Yield (keys, values, state1 | None, state2 | None) numpy pieces of at most chunk_pairs pairs."""
    meta = read_meta(path)
    n, dim = meta["n"], meta["dim"]
    if os.path.getsize(os.path.join(path, "keys.i64")) != 8 * n:
        raise ValueError(f"{path}: keys.i64 does not hold the {n} keys meta.json promises")
    kf = open(os.path.join(path, "keys.i64"), "rb")
    pfs = [open(os.path.join(path, p + ".f32"), "rb") for p in meta["planes"]]
    try:
        for s in range(0, n, chunk_pairs):
            m = min(chunk_pairs, n - s)
            keys = np.fromfile(kf, dtype="<i8", count=m)
            rows = [np.fromfile(f, dtype="<f4", count=m * dim).reshape(m, dim) for f in pfs]
            if keys.size != m or any(r.shape[0] != m for r in rows):
                raise ValueError(f"{path}: truncated checkpoint")
            yield (keys, *rows, *([None] * (3 - len(rows))))
    finally:
        kf.close()
        for f in pfs:
            f.close()


def load_into(table, path: str, chunk_pairs: int | None = None, keep=None) -> int:
    """Load `path` into `table`; keep(keys_tensor) -> bool mask restricts what is loaded (re-sharding)."""
    meta = read_meta(path)
    if meta["dim"] != table.dim:
        raise ValueError(f"checkpoint dim {meta['dim']} != table dim {table.dim}")
    chunk_pairs = chunk_pairs or int(getattr(table, "max_batch", 1 << 20))
    dev = getattr(table, "device", torch.device("cpu"))
    same_opt = int(meta["optimizer"]) == int(table.optimizer)   # state planes only mean something to the same optimizer
    loaded = 0
    for keys, vals, s1, s2 in iter_pairs(path, chunk_pairs):
        k = torch.from_numpy(keys).to(dev)
        parts = [torch.from_numpy(x).to(dev) if x is not None and (i == 0 or same_opt) else None for i, x in enumerate((vals, s1, s2))]
        if keep is not None:
            idx = torch.nonzero(keep(k)).view(-1)
            k = k[idx]
            parts = [p[idx] if p is not None else None for p in parts]
        if k.numel():
            table.import_(k, *parts)
            loaded += k.numel()
    return loaded


def save_sharded(sharded, path: str, chunk_slots: int = 1 << 22) -> int:
    """Every rank writes its shard under path/shard-<rank>-of-<world>; returns this rank's pair count."""
    d = os.path.join(path, f"shard-{sharded.rank:05d}-of-{sharded.world:05d}")
    return save_table(sharded.local, d, chunk_slots, extra={"rank": sharded.rank, "world": sharded.world})


def load_sharded(sharded, path: str, owner_of, chunk_pairs: int | None = None) -> int:
    """Load a sharded checkpoint written by ANY world size: this rank reads every shard directory and keeps the pairs
    it owns under the CURRENT world size (owner_of(keys) -> owner ranks); when the world size is unchanged it reads only
    its own directory.  Returns the pairs this rank loaded."""
    dirs = sorted(x for x in os.listdir(path) if x.startswith("shard-"))
    if not dirs:
        raise ValueError(f"{path}: no shard-* directories")
    mine = f"shard-{sharded.rank:05d}-of-{sharded.world:05d}"
    if mine in dirs and all(x.endswith(f"-of-{sharded.world:05d}") for x in dirs) and len(dirs) == sharded.world:
        return load_into(sharded.local, os.path.join(path, mine), chunk_pairs)
    n = 0
    for x in dirs:
        n += load_into(sharded.local, os.path.join(path, x), chunk_pairs, keep=lambda k: owner_of(k) == sharded.rank)
    return n

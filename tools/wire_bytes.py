"""Bytes a rank puts on its busiest xGMI link per sharded TRAINING step (lookup + backward), with and without the pre-exchange reductions
(MEE_SHARDED_DEDUP: distinct keys out / rows back for the lookup, ONE summed gradient row per distinct key for the backward) — computed on the
CPU from the bench's own key streams (bench.lookup_batches' Zipf(1.05) inverse-CDF sampler and its uniform stream, restated in numpy) and the
library's owner function (oracle.hash_batch).  Per position or distinct key: lookup = 8 B key out + 4 dim B row + 1 found byte back; backward =
8 B key + 4 dim B gradient row out.  usage: wire_bytes.py [OUT.md]  (no GPU needed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from meepoembedding_amd import synth

N_KEYS = 125_000_000          # per GPU; the table of G GPUs holds G x this (bench.py N > 1: weak scaling)


def batch(dist, n, n_keys, rng):
    if dist == "zipf":
        a = 1.05
        u = rng.random(n)
        hi = float(n_keys) ** (1 - a)
        idx = np.clip(np.floor((1 + u * (hi - 1)) ** (1 / (1 - a))).astype(np.int64) - 1, 0, n_keys - 1)
    else:
        idx = rng.integers(0, n_keys, n)
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(synth._GOLDEN) + np.uint64(1)
    return synth.mix64_np(z).view(np.int64)


def main():
    lines = ["# Bytes on a rank's busiest xGMI link per sharded training step: every occurrence vs one pair per distinct key (round 5)", "",
             "Computed by `tools/wire_bytes.py` (CPU, numpy) from the bench's key streams over G x 125M keys and the library's owner function; per rank B keys per step.",
             "`occ` = positions this rank sends to its busiest peer, `U` = distinct keys among them.  Lookup: 8 B out + (4 dim + 1) B back per unit;",
             "backward: (8 + 4 dim) B out per unit.  Without MEE_SHARDED_DEDUP the unit is a position (bytes follow B), with it a distinct key (bytes follow U):",
             "`mee_dedup_keys` in front of the lookup's exchange, `mee_dedup_sum` (fp64 row sums, rounded once) in front of the backward's.", "",
             "| stream | GPUs | B per rank | dim | occ to busiest peer | U to busiest peer | U / occ | lookup MB plain -> dedup | backward MB plain -> aggregated | step MB plain -> dedup | link time @ 76.8 GB/s, ms plain -> dedup |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    rng = np.random.default_rng(4)
    for dist in ("zipf", "uniform"):
        for G in (2, 4, 8):
            for B in (131072, 1 << 20):
                keys = batch(dist, B, N_KEYS * G, rng)
                owner = oracle.hash_batch(keys, 1, G)[2]
                me = 0
                occ = np.bincount(owner, minlength=G)
                uk = np.unique(keys)
                uown = oracle.hash_batch(uk, 1, G)[2]
                uni = np.bincount(uown, minlength=G)
                peers = [p for p in range(G) if p != me]
                bo, bu = max(occ[p] for p in peers), max(uni[p] for p in peers)
                for dim in (64, 128):
                    lk = lambda m: m * (8 + 4 * dim + 1) / 1e6
                    bw = lambda m: m * (8 + 4 * dim) / 1e6
                    plain, dd = lk(bo) + bw(bo), lk(bu) + bw(bu)
                    # the lookup's bytes travel in two directions (keys out, rows back); the busiest DIRECTION of the step carries rows back + nothing | keys + grads out
                    out_plain, out_dd = bo * (8 + 8 + 4 * dim) / 1e6, bu * (8 + 8 + 4 * dim) / 1e6     # direction rank -> peer: lookup keys + backward pairs
                    lines.append(f"| {dist} | {G} | {B} | {dim} | {bo} | {bu} | {bu / bo:.3f} | {lk(bo):.2f} -> {lk(bu):.2f} | {bw(bo):.2f} -> {bw(bu):.2f} | {plain:.2f} -> {dd:.2f} | "
                                 f"{out_plain / 76.8:.3f} -> {out_dd / 76.8:.3f} |")
    lines += ["", "The last column prices the direction rank -> busiest peer (lookup keys + backward pairs) at the nominal 76.8 GB/s per link direction (SURVEY.md §8e); the rows",
              "coming back load the opposite direction with 4 dim + 1 B per unit.  Uniform streams have U / occ ~ 1: the reductions buy nothing there and cost their local",
              "work (mee_dedup_sum: `profiles/r05_dedup.md`), which is why they are a flag of the context and not the default."]
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()

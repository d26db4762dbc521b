"""Randomised check of mee_dedup_sum against a torch reference (fp64 index_add) over streams that exercise every path of the kernel: uniform, Zipf with exponents 0.9-1.4,
a rotating set of hot keys of 1..60 000 occurrences (keys without a bucket of their own yet: long runs handed over, runs beyond 1024 sources in pieces, buckets beyond the LDS
list), single-key batches, small batches; dims 16 / 40 / 64 / 128; keys + counts + inverse only; other operators (dedup_keys, an apply over the distinct keys) in between.
usage: fuzz_dedup_sum.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth


def run(rounds, seed, quiet=False):
    """returns the number of batches checked (raises AssertionError on the first mismatch)"""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    EMPTY = -(1 << 63)
    pool = synth.keys_np(900, 0, 3_000_000)
    checked = 0
    for r in range(rounds):
        dim = int(rng.choice([16, 40, 64, 128]))
        n = int(rng.choice([3000, 70_000, 260_000, 600_000, 1_048_576]))
        if dim == 128:
            n = min(n, 600_000)
        with_opt = bool(rng.integers(0, 2)) and n <= 600_000
        t = LookupTable(1 << 16, dim, device=dev, max_batch=n, **({"optimizer": OPT_ADAGRAD} if with_opt else {}))
        kind = rng.choice(["uniform", "zipf", "hot_mix", "one_key", "two_phase"])
        a = float(rng.uniform(0.9, 1.4))
        for step in range(int(rng.integers(2, 6))):
            if kind == "uniform":
                keys = pool[rng.integers(0, pool.size, n)]
            elif kind == "zipf" or (kind == "two_phase" and step < 2):
                keys = pool[(rng.zipf(1.0 + max(a - 1.0, 0.02), n) - 1) % pool.size] if a > 1.0 else pool[np.minimum((rng.pareto(a, n) * 3).astype(np.int64), pool.size - 1)]
            elif kind == "one_key":
                keys = np.full(n, pool[step], np.int64)
            else:   # hot keys that were not hot in the batch before: long runs in hash buckets, in every size class
                m = int(rng.integers(5, 200))
                hot = pool[(np.arange(m) + 977 * step + 13 * r) % 50_000]
                reps = np.minimum(rng.integers(1, 3000, size=m) ** int(rng.integers(1, 3)) // int(rng.choice([1, 7, 300])) + 1, n // 4)
                body = np.repeat(hot, reps)[: n // 2]
                keys = np.concatenate([body, pool[60_000 + rng.integers(0, pool.size - 60_000, n - body.size)]])
                rng.shuffle(keys)
            if rng.integers(0, 3) == 0:
                keys = keys.copy(); keys[rng.integers(0, n, 5)] = EMPTY
            kt = torch.from_numpy(np.ascontiguousarray(keys)).to(dev)
            g = torch.randn((n, dim), device=dev)
            rows_mode = int(rng.integers(0, 4)) != 0
            if rng.integers(0, 2):
                u0, i0 = t.dedup_keys(kt)
                v = kt != EMPTY
                assert torch.equal(u0[i0[v]], kt[v]), f"round {r} step {step}: dedup_keys"
            uniq, gs, cnt, inv = t.dedup_sum(kt, g if rows_mode else None, miss_index=-1)
            valid = kt != EMPTY
            assert torch.equal(uniq[inv[valid]], kt[valid]) and bool((inv[~valid] == -1).all()), f"round {r} step {step} ({kind}, n {n}, dim {dim}): inverse"
            ref_cnt = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, inv[valid], torch.ones(int(valid.sum()), dtype=torch.int64, device=dev))
            assert torch.equal(cnt.long(), ref_cnt), f"round {r} step {step} ({kind}, n {n}, dim {dim}): counts"
            live = cnt > 0
            assert int(live.sum()) == int(torch.unique(kt[valid]).numel()) and bool((uniq[~live] == EMPTY).all()), f"round {r} step {step}: distinct keys / padding"
            if rows_mode:
                ref = torch.zeros((n, dim), dtype=torch.float64, device=dev).index_add_(0, inv[valid], g[valid].double())
                err = (gs[live].double() - ref[live]).abs().max().item()
                scale = ref[live].abs().max().item()
                assert err <= 1e-6 * max(scale, 1.0) + 1e-9, f"round {r} step {step} ({kind}, n {n}, dim {dim}): summed rows off by {err} (scale {scale})"
                once = live & (cnt == 1)
                src = torch.full((n,), -1, dtype=torch.int64, device=dev); src[inv[valid]] = torch.nonzero(valid).flatten()
                assert torch.equal(gs[once], g[src[once]]), f"round {r} step {step}: single occurrences must be copies"
                if with_opt and rng.integers(0, 2):
                    t.apply_adagrad(uniq, gs, lr=0.01)   # (keys absent from the table: skipped; what matters is the operator in between)
            checked += 1
        assert t.status() == 0, f"round {r}: status {t.status()}"
        del t
        if not quiet: print(f"round {r}: {kind} a={a:.2f} n={n} dim={dim} ok", flush=True)
    print(f"fuzz_dedup_sum ok: {checked} batches")
    return checked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

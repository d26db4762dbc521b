"""Randomised check of the sparse-Adagrad apply against a torch reference (fp64 gradient sums rounded once, the update in fp32 as SPEC.md §4 writes it) on streams that
change their skew from batch to batch — so that every plan meets batches it was not made for: uniform batches behind skewed ones and the reverse, giant keys that share a
hash bucket on the first skewed batch (the LEAN kernel's split-bucket path: by position, crowded, by key), hundreds of mid-sized keys (more than the hot-key set holds),
single-key batches; probing and located (behind the training forward) applies, host in step or running ahead.  usage: fuzz_apply.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth


def run(rounds, seed, quiet=False):
    """returns the number of batches checked (raises AssertionError on the first mismatch)"""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    lr, eps, acc0 = 0.05, 1e-10, 0.1
    checked = 0
    for r in range(rounds):
        dim = int(rng.choice([16, 32, 64, 40]))
        n_keys = int(rng.choice([60_000, 400_000, 1_500_000]))
        n = int(rng.choice([20_000, 120_000, 262_144, 500_000]))
        keys = synth.keys_np(700 + r, 0, n_keys)
        kt_all = torch.from_numpy(keys).to(dev)
        t = LookupTable(int(n_keys / 0.7), dim, device=dev, max_batch=max(n, 500_000), optimizer=OPT_ADAGRAD, initial_accumulator=acc0)
        w = torch.randn((n_keys, dim), device=dev) * 0.1
        for s0 in range(0, n_keys, 500_000):
            t.insert(kt_all[s0:s0 + 500_000], w[s0:s0 + 500_000].contiguous())
        acc = torch.full((n_keys, dim), acc0, device=dev)
        host_in_step = bool(rng.integers(0, 2))
        for step in range(int(rng.integers(3, 7))):
            kind = rng.choice(["uniform", "zipf", "giants", "mids", "one_key"])
            if kind == "uniform":
                idx = rng.integers(0, n_keys, n)
            elif kind == "zipf":
                idx = (rng.zipf(float(rng.uniform(1.02, 1.5)), n) - 1) % n_keys
            elif kind == "one_key":
                idx = np.full(n, int(rng.integers(0, n_keys)))
            else:
                m = int(rng.integers(20, 90)) if kind == "giants" else int(rng.integers(150, 400))
                hot = rng.integers(0, n_keys, m)
                reps = rng.integers(700, 6000, size=m) if kind == "giants" else rng.integers(100, 700, size=m)
                body = np.repeat(hot, reps)[: n * 3 // 4]
                idx = np.concatenate([body, rng.integers(0, n_keys, n - body.size)])
                rng.shuffle(idx)
            it = torch.from_numpy(idx.astype(np.int64)).to(dev)
            kb = kt_all[it]
            g = torch.randn((n, dim), device=dev) * 0.02
            if rng.integers(0, 2):
                out, found, slots = t.find_located(kb, prepare_apply=True)
                assert bool(found.all())
                t.apply_adagrad(kb, g, lr=lr, eps=eps, slots=slots)
            else:
                t.apply_adagrad(kb, g, lr=lr, eps=eps)
            if host_in_step:
                torch.cuda.synchronize()
            # reference: one update per distinct key from the fp64 sum of its rows, rounded once
            gs = torch.zeros((n_keys, dim), dtype=torch.float64, device=dev).index_add_(0, it, g.double()).float()
            touched = torch.zeros(n_keys, dtype=torch.bool, device=dev); touched[it] = True
            a2 = torch.addcmul(acc, gs, gs)                       # fmaf(g, g, acc)
            q = gs / (a2.sqrt() + eps)
            w2 = torch.addcmul(w, q, torch.full_like(q, -lr))     # fmaf(-lr, q, w)
            acc = torch.where(touched[:, None], a2, acc); w = torch.where(touched[:, None], w2, w)
            if step % 2 == 1 or kind in ("giants", "mids"):
                got, f = t.find(kt_all[: min(n_keys, 1 << 20)])
                ref = w[: got.shape[0]]
                err = (got - ref).abs().max().item()
                assert bool(f.all()) and err <= 2e-6 * max(ref.abs().max().item(), 1.0) + 1e-8, f"round {r} step {step} ({kind}, n {n}, keys {n_keys}, dim {dim}): rows off by {err}"
                checked += 1
        assert t.status() == 0, f"round {r}: status {t.status()}"
        del t
        if not quiet: print(f"round {r}: keys {n_keys} n {n} dim {dim} host_in_step {host_in_step} ok", flush=True)
    print(f"fuzz_apply ok: {checked} checked batches")
    return checked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

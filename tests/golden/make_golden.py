"""Generates the committed golden fixtures (run in the build container; needs torch CPU).

  hash_kat.json          SPEC §1 known answers.  Produced by oracle/pyspec.py (pure Python ints) and checked
                         here against oracle/meepo_oracle.c before writing: two independent implementations.
  optimizer_golden.npz   SPEC §4 cross-check against THIRD-PARTY math: torch.optim.Adagrad and
                         torch.optim.SparseAdam on nn.Embedding(sparse=True) (duplicate indices coalesced by
                         sum), 5 steps, dim 16 and 64; round 4: five more cases (dim 128, non-default eps and betas,
                         initial_accumulator_value 0 and 0.1).  torch is an installed package, not reference code.

The reference snapshot (/root/reference) has no tests, fixtures or code to generate vectors from, so these are
the only pins the oracle has ("parity unpinned" by upstream).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402
from oracle import pyspec  # noqa: E402


def hash_kat():
    edge = [0, 1, -1, 2, -2, (1 << 63) - 1, -(1 << 63) + 1, -(1 << 63) + 2, 1 << 32, (1 << 32) - 1, 0x0123456789ABCDEF,
            -0x0123456789ABCDEF, 42, 1 << 62, -(1 << 62)]
    keys = edge + [pyspec.splitmix64_stream(7, i) for i in range(1000 - len(edge))]
    nbs = [1, 7, 16, 625, 65536, 8388608, (1 << 40) + 12345]
    gs = [1, 2, 3, 4, 8, 64]
    kat = {"keys": [str(k) for k in keys],
           "mix64": [str(pyspec.mix64(pyspec.u64(k))) for k in keys],
           "mix64b": [str(pyspec.mix64b(pyspec.u64(k))) for k in keys],
           "bucket": {str(nb): [str(pyspec.bucket(k, nb)) for k in keys] for nb in nbs},
           "owner": {str(g): [pyspec.owner(k, g) for k in keys] for g in gs},
           "step": {str(nb): [str(pyspec.step(k, nb)) for k in keys] for nb in nbs},
           "initial_row": {"seed": 11, "scale": 0.05, "dim": 8,
                           "rows": [[float(x) for x in pyspec.initial_row(k, 8, 0.05, 11)] for k in keys[:64]]}}
    # second implementation must agree before anything is written
    ka = np.array(keys, dtype=np.int64)
    for nb in nbs:
        for g in gs:
            mix, bkt, own = oracle.hash_batch(ka, nb, g)
            assert [str(int(x)) for x in mix] == kat["mix64"]
            assert [str(int(x)) for x in bkt] == kat["bucket"][str(nb)]
            assert [int(x) for x in own] == kat["owner"][str(g)]
        assert [str(oracle.lib().meo_step(int(k), nb)) for k in keys] == kat["step"][str(nb)]
    t = oracle.OracleTable(64, 8, initializer=oracle.INIT_UNIFORM, init_scale=0.05, init_seed=11)
    for k, row in zip(keys[:64], kat["initial_row"]["rows"]):
        assert np.array_equal(t.initial_row(k), np.array(row, np.float32)), k
    with open(os.path.join(HERE, "hash_kat.json"), "w") as f:
        json.dump(kat, f)
    print("hash_kat.json:", len(keys), "keys")


def optimizer_golden():
    out = {}
    for dim in (16, 64):
        g = torch.Generator().manual_seed(1234 + dim)
        rows, steps, batch = 257, 5, 300
        w0 = torch.rand(rows, dim, generator=g) - 0.5
        idx = torch.randint(0, rows, (steps, batch), generator=g)
        idx[:, :40] = idx[:, 40:80]  # force duplicates inside every batch
        grads = torch.randn(steps, batch, dim, generator=g) * 0.1
        out[f"w0_{dim}"], out[f"idx_{dim}"], out[f"grads_{dim}"] = w0.numpy(), idx.numpy(), grads.numpy()
        for name in ("adagrad", "adam"):
            emb = torch.nn.Embedding(rows, dim, sparse=True)
            with torch.no_grad():
                emb.weight.copy_(w0)
            if name == "adagrad":
                opt = torch.optim.Adagrad(emb.parameters(), lr=0.05, eps=1e-10, initial_accumulator_value=0.1)
            else:
                opt = torch.optim.SparseAdam(emb.parameters(), lr=0.01, betas=(0.9, 0.999), eps=1e-8)
            for s in range(steps):
                opt.zero_grad()
                y = emb(idx[s])
                y.backward(grads[s])
                opt.step()
            out[f"{name}_w_{dim}"] = emb.weight.detach().numpy().copy()
    # round 4: more of the parameter space — dim 128 (configs[3]'s row width), non-default eps / betas, initial_accumulator_value 0 and 0.1.
    # Each case is self-describing (cases_json) so that the tests iterate over whatever is here.
    cases = [dict(tag="adagrad_d128", opt="adagrad", dim=128, lr=0.05, eps=1e-10, acc0=0.1),
             dict(tag="adagrad_d64_eps1e-6_acc0", opt="adagrad", dim=64, lr=0.02, eps=1e-6, acc0=0.0),
             dict(tag="adagrad_d128_eps1e-4_acc0", opt="adagrad", dim=128, lr=0.1, eps=1e-4, acc0=0.0),
             dict(tag="adam_d128", opt="adam", dim=128, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8),
             dict(tag="adam_d64_eps1e-6_b0.8_0.99", opt="adam", dim=64, lr=0.003, beta1=0.8, beta2=0.99, eps=1e-6)]
    for ci, c in enumerate(cases):
        g = torch.Generator().manual_seed(4321 + ci)
        rows, steps, batch, dim = 193, 6, 160, c["dim"]
        w0 = torch.rand(rows, dim, generator=g) - 0.5
        # every batch: 112 distinct rows, 48 of them twice — PAIRS only, never more: torch coalesces duplicates with an fp32 sum, which equals
        # SPEC.md's "fp64 sum, rounded once" bit for bit for two terms; with three or more terms torch's own rounding error, amplified where the
        # terms cancel (and again by eps when the accumulator starts at 0), exceeds the 1e-6 the comparison allows — longer runs are checked
        # against the oracle, not against torch
        idx = torch.stack([torch.cat([p[:112], p[:48]]) for p in (torch.randperm(rows, generator=g) for _ in range(steps))])
        grads = torch.randn(steps, batch, dim, generator=g) * 0.1
        emb = torch.nn.Embedding(rows, dim, sparse=True)
        with torch.no_grad():
            emb.weight.copy_(w0)
        if c["opt"] == "adagrad":
            opt = torch.optim.Adagrad(emb.parameters(), lr=c["lr"], eps=c["eps"], initial_accumulator_value=c["acc0"])
        else:
            opt = torch.optim.SparseAdam(emb.parameters(), lr=c["lr"], betas=(c["beta1"], c["beta2"]), eps=c["eps"])
        for s in range(steps):
            opt.zero_grad()
            emb(idx[s]).backward(grads[s])
            opt.step()
        t = c["tag"]
        out[f"c_{t}_w0"], out[f"c_{t}_idx"], out[f"c_{t}_grads"], out[f"c_{t}_w"] = w0.numpy(), idx.numpy(), grads.numpy(), emb.weight.detach().numpy().copy()
    out["cases_json"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "optimizer_golden.npz"), **out)
    print("optimizer_golden.npz written")


if __name__ == "__main__":
    if "--optimizer-only" not in sys.argv:
        hash_kat()
    optimizer_golden()

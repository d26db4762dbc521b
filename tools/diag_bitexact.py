import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from meepoembedding_amd import *
from meepoembedding_amd import synth
dev = torch.device('cuda', 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dim, n = 64, 10000
keys = synth.keys_np(5, 0, n); rows = synth.rows_np(keys, dim, 2)
g = (synth.rows_np(keys, dim, 6) * 0.02).astype(np.float32)
for kind, okind in ((OPT_ADAGRAD, oracle.OPT_ADAGRAD), (OPT_ADAM, oracle.OPT_ADAM)):
    t = LookupTable(16384, dim, device=dev, optimizer=kind, max_batch=n); o = oracle.OracleTable(16384, dim, optimizer=okind)
    t.insert(T(keys), T(rows)); o.insert(keys, rows)
    for s in range(3):
        if kind == OPT_ADAGRAD:
            t.apply_adagrad(T(keys), T(g), lr=0.01); o.apply_adagrad(keys, g, 0.01, 1e-10)
        else:
            t.apply_adam(T(keys), T(g), lr=0.001, step=s + 1); o.apply_adam(keys, g, 0.001, 0.9, 0.999, 1e-8, s + 1)
        ge = [x.cpu().numpy() if x is not None else None for x in t.export(with_state=True)]
        oe = o.export(with_state=True)
        a, b = np.argsort(ge[0]), np.argsort(oe[0])
        for nm, x, y in zip(("w", "s1", "s2"), ge[1:], oe[1:]):
            if y is None: continue
            x, y = x[a], y[b]
            d = x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64)
            print(kind, "step", s, nm, "mismatch", int((d != 0).sum()), "of", d.size, "max ulp", int(np.abs(d).max()))

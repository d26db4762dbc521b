"""Random operator sequences against the ASan+UBSan build of the CPU oracle (oracle/Makefile target `asan`).
Run as: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_oracle.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
# point the binding at the sanitizer build
oracle._LIB_PATH = os.path.join(os.path.dirname(oracle.__file__), 'libmeepo_oracle_asan.so')
oracle._lib = None
orig_build = oracle.build
oracle.build = lambda force=False: oracle._LIB_PATH
import numpy as np
from meepoembedding_amd import synth
rng = np.random.default_rng(0)
for opt in (oracle.OPT_ADAGRAD, oracle.OPT_ADAM):
    t = oracle.OracleTable(16 * 40, 8, optimizer=opt, initializer=oracle.INIT_UNIFORM, init_scale=0.1, init_seed=1)
    uni = synth.keys_np(3, 0, 900)
    for it in range(300):
        n = int(rng.integers(0, 400)); keys = uni[rng.integers(0, 900, n)]
        if n and rng.random() < 0.1: keys[0] = oracle.EMPTY_KEY
        rows = rng.standard_normal((n, 8)).astype(np.float32)
        op = rng.integers(0, 9)
        if op == 0: t.insert(keys, rows)
        elif op == 1: t.assign(keys, rows)
        elif op == 2: t.remove(keys[: n // 3])
        elif op == 3: t.find(keys, threads=3)
        elif op == 4: t.find_or_insert(keys)
        elif op == 5: (t.apply_adagrad(keys, rows, 0.01, 1e-10) if opt == oracle.OPT_ADAGRAD else t.apply_adam(keys, rows, 0.01, 0.9, 0.999, 1e-8, it + 1))
        elif op == 6: oracle.dedup_sum(keys, rows, 8)
        elif op == 7: oracle.partition(keys, 5)
        else: t.export(with_state=True); t.find_plane(1, keys); t.assign_plane(1, keys, rows)
    t.close()
print("asan/ubsan run clean")

"""GPU: run the plain C++ host program (tests/cabi/cabi_test.cpp) that talks to libmeepo_hip.so through the C-ABI only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cabi_test", "host_cpp_test", "threads_test"])
def test_cpp_host_programs(dev, name):
    """cabi_test: raw C-ABI from C++.  host_cpp_test: the C++ host layer (include/meepo_embedding.hpp) — Table,
    TieredTable with a pinned-host cold half, Router + PeerExchange sharded pipeline.  threads_test: 4 host threads, each with its own stream,
    200 mee_find_ex calls each with different per-call cache-policy flags on ONE table — results equal to a serial pass, mee_last_error()
    thread-local (one thread provokes errors, the others' slots stay empty)."""
    exe = os.path.join(ROOT, "build", name)
    assert os.path.exists(exe), f"build/{name} missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{name} ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,threads", [(2, False), (4, False), (5, False), (8, True), (3, True)], ids=["2", "4", "5", "8-threads", "3-threads"])
def test_sharded_ranks_from_plain_cpp(dev, ranks, threads):
    """tests/cabi/sharded_mp_test.cpp: the program forks `ranks` processes BEFORE any HIP call and drives insert -> size -> find -> sparse Adagrad
    -> find -> remove through mee_sharded_* (exact segments, then padded segments + pre-exchange dedup), checking key-derived rows and the
    update computed on the host.  With fewer GPUs than ranks the library binds the shared-memory stand-in for librccl (several ranks on one
    device); with enough GPUs it binds RCCL itself and the ranks exchange over xGMI.  `threads`: the ranks are threads of ONE process on device 0 — that is how
    G = 8 (the node's GPU count: segment bookkeeping, padded capacities, the skewed step's pre-exchange aggregation) runs on a box that admits at most 6
    processes per job on its GPU (this test runner is one of them); always over the stand-in."""
    import torch
    exe = os.path.join(ROOT, "build", "sharded_mp_test")
    assert os.path.exists(exe), "build/sharded_mp_test missing: run __graft_entry__.build()"
    env = dict(os.environ)
    if threads or torch.cuda.device_count() < ranks:
        env["MEE_RCCL_LIB"] = os.path.join(ROOT, "build", "libfake_rccl.so")
    r = subprocess.run([exe, str(ranks)] + (["threads"] if threads else []), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sharded_mp_test ok" in r.stdout

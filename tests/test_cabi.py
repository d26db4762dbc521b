"""GPU: run the plain C++ host program (tests/cabi/cabi_test.cpp) that talks to libmeepo_hip.so through the C-ABI only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cabi_test", "host_cpp_test"])
def test_cpp_host_programs(dev, name):
    """cabi_test: raw C-ABI from C++.  host_cpp_test: the C++ host layer (include/meepo_embedding.hpp) — Table,
    TieredTable with a pinned-host cold half, Router + PeerExchange sharded pipeline."""
    exe = os.path.join(ROOT, "build", name)
    assert os.path.exists(exe), f"build/{name} missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{name} ok" in r.stdout

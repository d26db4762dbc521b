"""GPU: run the plain C++ host program (tests/cabi/cabi_test.cpp) that talks to libmeepo_hip.so through the C-ABI only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cabi_cpp_host(dev):
    exe = os.path.join(ROOT, "build", "cabi_test")
    assert os.path.exists(exe), "build/cabi_test missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_test ok" in r.stdout

"""Short runs of the randomised checks under tools/ (fuzz_dedup_sum.py, fuzz_apply.py) inside the GPU suite: streams that change their skew from batch to batch, every
path of the dedup and apply kernels, against torch references.  The long runs of the round are under profiles/ (r05_fuzz_*.txt)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_dedup_sum(dev, seed):
    assert _tool("fuzz_dedup_sum").run(3, seed, quiet=True) >= 6


@pytest.mark.parametrize("seed", [201, 202])
def test_fuzz_apply(dev, seed):
    assert _tool("fuzz_apply").run(3, seed, quiet=True) >= 3

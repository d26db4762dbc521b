"""CPU: the oracle (the checker everything else is judged by) must itself be memory-clean — random operator sequences
against its ASan + UBSan build (SURVEY.md §5 'race detection / sanitizers' row; GPU sanitizers are not available)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(built):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_oracle.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "run clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


import pytest


@pytest.mark.parametrize("sanitizer", ["address", "thread"])
def test_cabi_host_side_under_sanitizers(built, sanitizer):
    """SURVEY.md §5 'sanitizers', the C-ABI's HOST code (GPU sanitizers are not available on the pool): the host halves of csrc/*.hip (hipcc
    --cuda-host-only) are linked against a HIP stand-in — device memory is host memory, copies are memcpy, launches do nothing
    (tests/cabi/hip_host_stub.cpp) — and the shared-memory stand-in for librccl, everything instrumented with ASan / TSan, and EIGHT rank threads
    (the node's GPU count) drive contexts of every flavour (exact, padded, dedup, hot/cold pair), every mee_sharded_* operator, the single-table
    operators, routers and peer contexts through it (tests/cabi/sanitize_host.cpp).  No value is checked — kernels do not run, this pins nothing
    about parity —: the sanitizers watch the library's own bookkeeping (the bounds of every copy, lifetimes, the process-wide state rank threads
    share)."""
    out = os.path.join(ROOT, "build", f"san_{sanitizer}")
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cabi", "build_sanitized.sh"), sanitizer, out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, MEE_RCCL_LIB=os.path.join(out, "libfake_rccl.so"), MEE_FAKE_RCCL_SLOT_MB="8", ASAN_OPTIONS="detect_leaks=0",
               TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(out, "sanitize_host"), "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitize_host ok" in r.stdout and "Sanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-6000:]

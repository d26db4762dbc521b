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

"""GPU: bench.py's output contract — exactly one JSON line on stdout with the required fields — at N=1 (small table) and,
as a rehearsal of the N>1 flow, with two ranks sharing this box's GPU over gloo (populate by owner, transport probe)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _one_json_line(cmd):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must be ONE line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_gpu_contract(dev):
    res = _one_json_line([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--keys", "2000000", "--no-extras"])
    assert REQUIRED <= set(res) and "cpu_baseline" in res
    assert res["n_gpus"] == 1 and res["steps"] == 20 and res["warmup"] == 5 and res["vs_baseline"] is None and res["value"] > 1e8
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert "workload" in res["config"] and "model" not in res["config"] and "launch" in res["config"]
    assert rf["avg_launch_us"] > 0 and "window" in rf and "traffic_source" in rf
    st = res["streams"]   # uniform / Zipf / 90-10 hit-miss, probe length, read-only GB/s, launch-size sweep
    assert {"uniform", "zipf_1.05", "hit90_miss10", "uniform_with_streaming_load_hint", "launch_size_sweep"} <= set(st), st
    for name in ("uniform", "zipf_1.05", "hit90_miss10"):
        assert st[name]["lookups_per_s"] > 0 and st[name]["mean_probe_length_buckets"] >= 0.9 and st[name]["read_only_GBps"] > 0
    assert st["hit90_miss10"]["mean_probe_length_buckets"] > st["zipf_1.05"]["mean_probe_length_buckets"] - 1e-9
    assert len(st["launch_size_sweep"]["us_per_launch"]) == 6 and st["launch_size_sweep"]["fit_us_per_262144_lookups"] > 0


@pytest.mark.gpu
def test_bench_two_rank_rehearsal(dev):
    res = _one_json_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--keys", "2000000", "--batch", "65536",
                          "--steps", "5", "--warmup", "2"])
    assert REQUIRED <= set(res) and res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert "transport" in res["config"]["workload"]


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_native_transport(dev):
    """The N=2 flow through the exchange behind the C-ABI (self-test children, both segment layouts probed against the
    torch.distributed path, the timed steps): two ranks share this box's GPU, so the library binds the shared-memory stand-in for
    RCCL (tests/cabi/fake_rccl.cpp).  Timings mean nothing here; the flow and the result checks are what is rehearsed."""
    env = dict(os.environ, MEE_RCCL_LIB=os.path.join(ROOT, "build", "libfake_rccl.so"), MEE_FAKE_RCCL_SLOT_MB="32")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--transport", "native", "--keys", "2000000",
                        "--batch", "65536", "--steps", "5", "--warmup", "2", "--verbose"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert REQUIRED <= set(res) and res["n_gpus"] == 2 and res["value"] > 0
    assert "behind the C-ABI" in res["config"]["workload"], (res["config"]["workload"], r.stderr[-2000:])
    # the same run with the transport's post-run check forced to fail: every rank falls back to the torch.distributed path, the timed
    # region is run again, and the line says so — an automated N-GPU sweep must get its line either way
    env["MEE_BENCH_FAIL_TRANSPORT_CHECK"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--transport", "native", "--no-selftest",
                        "--mode", "train", "--keys", "2000000", "--batch", "65536", "--steps", "5", "--warmup", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "fallback" in res["config"]["workload"], res["config"]["workload"]

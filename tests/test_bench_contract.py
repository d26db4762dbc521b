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
    assert st["north_star_batch_1M"]["lookups_per_launch"] == 1 << 20 and st["north_star_batch_1M"]["default_policy"]["frac_of_hbm_roofline"] > 0
    # the headline rotates its results over more output bytes than the Infinity Cache holds and says so; the reused-buffer figure is beside it
    assert rf["out_buffers"] >= 5 and rf["frac_out_rotating"] == rf["frac"] and rf["frac_out_reused"] > 0
    # ... with the caller's hints passed per call (mee_find_ex), and the no-hint figure (the library's own rule into the same rotating buffers) beside it
    # (since round 5 the library's own rule sees the rotation and streams its stores too: the two figures differ by the row-load hint and by noise)
    assert 0 < rf["frac_library_default"] <= rf["frac"] * 1.05 and "mee_find_ex_flags" in rf["out_store_policy"]
    assert "rotating" in res["config"]["workload"] and res["config"]["launch_comparison"]["eager_us_per_step"] > 0
    # configs[2] travels in the default line: find + sparse Adagrad step and the apply alone, uniform and Zipf(1.05)
    c2 = res["configs2"]
    for name in ("uniform", "zipf_1.05"):
        assert c2[name]["step"]["us"] > c2[name]["apply_alone"]["us"] > 0 and 0 < c2[name]["step"]["frac_of_hbm_roofline"] < 1
        # north_star "Adagrad/Adam": the same rows with sparse Adam
        assert c2[name]["step_adam"]["us"] > c2[name]["apply_alone_adam"]["us"] > 0 and 0 < c2[name]["step_adam"]["frac_of_hbm_roofline"] < 1
    assert c2["zipf_1.05"]["unique_keys_per_batch"] < c2["uniform"]["unique_keys_per_batch"]
    assert cb["table_keys"] == 2_000_000 and str(cb["cores"]) in cb["by_threads"] and cb["value"] == max(cb["by_threads"].values())


@pytest.mark.gpu
def test_bench_two_rank_rehearsal(dev):
    res = _one_json_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--keys", "2000000", "--batch", "65536",
                          "--steps", "5", "--warmup", "2"])
    assert REQUIRED <= set(res) and res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert "transport" in res["config"]["workload"]
    # SURVEY 8d config 4 (iii) in the one line: the uniform stream and Zipf(1.05) without and with pre-exchange dedup, each with its value, the
    # transport that carried it and the bytes on the busiest link
    st = res["streams"]
    for name in ("uniform", "zipf_1.05", "zipf_1.05_dedup"):
        row = st[name]
        assert row["value"] > 0 and row["ms_per_step"] > 0 and row["transport"] and row["xgmi"]["bytes_on_busiest_link"] > 0 and row["xgmi"]["frac"] > 0
    assert st["zipf_1.05_dedup"]["pre_exchange_dedup"] and not st["zipf_1.05"]["pre_exchange_dedup"]
    assert st["zipf_1.05_dedup"]["xgmi"]["bytes_on_busiest_link"] < st["zipf_1.05"]["xgmi"]["bytes_on_busiest_link"]   # only the distinct keys travel
    assert st["zipf_1.05"]["unique_fraction_per_rank_batch"] < st["uniform"]["unique_fraction_per_rank_batch"]


@pytest.mark.gpu
def test_bench_self_launches_its_ranks(dev):
    """`python bench.py --gpus 2` with NO launcher around it: the command starts its own two ranks (before it touches the GPU), relays rank 0's
    line and exits with the ranks' code — the form a driver uses for every N.  gloo backend: both ranks share this box's one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--keys", "2000000", "--batch", "65536", "--steps", "5", "--warmup", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:500]
    res = json.loads(lines[0])
    assert REQUIRED <= set(res) and res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    rf, xg = res["roofline"], res["xgmi"]
    assert rf["kernel"] == "find_kernel" and 0 < rf["frac"] < 1 and rf["lookups_per_launch"] > 0
    # two ranks, uniform keys: about half of a rank's 65536 keys go to the peer (8 B each), their rows (256 B + found byte) come back
    assert 0.8 < xg["bytes_on_busiest_link"] / (32768 * (8 + 257) * 1.0) < 1.6 and xg["frac"] > 0 and xg["link_peak_GBps_per_direction"] == 76.8
    # a rank count that the launcher cannot honour fails loudly, before anything is started
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--steps", "1", "--warmup", "0"], cwd=ROOT, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "GPUs" in (r.stderr + r.stdout)


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


def test_bench_host_probes_run_without_a_gpu():
    """The two probes bench.py makes BEFORE any HIP call — GPUs visible (sysfs, cut by *_VISIBLE_DEVICES) and host RAM this process may still take (MemAvailable cut by
    the cgroup limit) — are plain file reads: they must work (and stay sane) on a box without a GPU."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = bench.visible_gpu_count()
    assert isinstance(n, int) and 0 <= n <= 64
    os.environ["HIP_VISIBLE_DEVICES"] = "0"
    try:
        assert bench.visible_gpu_count() <= 1
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    ram = bench._host_ram_available_gb()
    assert 0.0 <= ram < 1e6

#!/usr/bin/env python3
"""bench.py — key-lookups/sec of the HIP find path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W        configs[1]: 100M keys, dim 64, forward find, 256K-key batches
    python bench.py --gpus N --steps K --warmup W        row-sharded: 125M keys per GPU (N=8: the 1B-key table), dim 64, 1M lookups per
                                                          rank per step, RCCL exchange of keys / rows.  Without a launcher around it the
                                                          command starts its own N ranks (one per GPU) before touching the GPU and relays
                                                          rank 0's line; under `python -m torch.distributed.run --nproc-per-node N … bench.py
                                                          --gpus N …` it is one of the ranks.

A step = one pass of the hot path over one batch of synthetic keys already resident in HBM.  N=1: one mee_find
launch, results rotating over 6 output buffers (384 MB: more than the Infinity Cache absorbs).  N>1: one sharded find
(partition, keys to their owners, local find, rows back, un-permute).  Rank 0 prints ONE JSON line.  `roofline` is for the
dominant kernel (find_kernel) from HIP events around back-to-back launches; N=1 also carries the SURVEY 8d stream table,
the configs[2] (find + sparse Adagrad) rows and `cpu_baseline` (the in-repo CPU oracle, "port": the reference snapshot has
no implementation) on the configuration the GPU number is quoted on; N>1 carries `xgmi` (bytes on the busiest link).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def algorithmic_bytes_per_lookup(dim: int) -> int:
    """batch key 8 + matched table key 8 + row read dim*4 + dense row write dim*4."""
    return 16 + 8 * dim


def populate(table, synth, n_keys, dim, dev, chunk, owner_rank=None, world=1, hash_batch=None, log=None):
    """Insert keys_t(seed=1, 0..n_keys) with rows_t(seed=2); sharded: only the keys this rank owns."""
    t0 = time.time()
    for s in range(0, n_keys, chunk):
        c = min(chunk, n_keys - s)
        k = synth.keys_t(1, s, c, dev)
        if owner_rank is not None and world > 1:
            own = hash_batch(k, 1, world)[2]
            k = k[own == owner_rank].contiguous()
        if k.numel():
            table.insert(k, synth.rows_t(k, dim, 2))
        if log and (s // chunk) % 50 == 0:
            log(f"populate {s + c}/{n_keys} keys ({time.time() - t0:.1f}s)")
    torch.cuda.synchronize(dev)


def lookup_batches(synth, n_keys, batch, n_batches, dist_name, dev, seed):
    """Pre-generate lookup batches over the inserted key set (uniform or Zipf(1.05) by key index)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    for _ in range(n_batches):
        if dist_name == "zipf":
            # inverse-CDF sampling of a truncated Zipf(alpha=1.05) over ranks 1..n_keys (continuous approximation)
            a = 1.05
            u = torch.rand(batch, device=dev, generator=g, dtype=torch.float64)
            hi = float(n_keys) ** (1 - a)
            r = (1 + u * (hi - 1)) ** (1 / (1 - a))
            idx = (r.floor().to(torch.int64) - 1).clamp_(0, n_keys - 1)
        else:
            idx = torch.randint(0, n_keys, (batch,), device=dev, generator=g)
        out.append(synth.mix64_t((idx + 1) * synth._s64(synth._GOLDEN) + synth._s64(1)))
    return out


def _physical_cores(allowed) -> int:
    """distinct (package, core) pairs among the CPUs this process may run on (sysfs topology); 0 if unreadable"""
    seen = set()
    try:
        for c in allowed:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
    except OSError:
        return 0
    return len(seen)


def _cpu_quota():
    """CPUs' worth of run time the cgroup grants this process (cgroup v2 cpu.max — the container's own file or, from /proc/self/cgroup, any
    ancestor's — or the v1 cfs quota), or None when unlimited / unreadable"""
    def v2(path):
        try:
            q, p = open(path).read().split()[:2]
            return None if q == "max" else float(q) / float(p)
        except (OSError, ValueError):
            return None
    best = v2("/sys/fs/cgroup/cpu.max")
    try:
        for ln in open("/proc/self/cgroup"):
            parts = ln.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                d = parts[2].rstrip("/")
                while d:
                    q = v2("/sys/fs/cgroup" + d + "/cpu.max")
                    best = q if (q is not None and (best is None or q < best)) else best
                    d = d.rsplit("/", 1)[0]
    except OSError:
        pass
    if best is not None:
        return best
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(synth, dim, batch, n_keys_config, load, budget_s=16.0):
    """In-repo CPU oracle find (persistent worker pool) on the configuration the GPU number is quoted on: configs[1]'s table (100M keys,
    dim 64, load 0.75: 35 GB of host DRAM) when the box has >= 64 GB available, else the largest table that fits (stated in `sample`);
    uniform 256K-key batches; 1 thread, one thread per physical core, every hardware thread.  `value` = the best of them."""
    import oracle
    allowed = sorted(os.sched_getaffinity(0))
    hw_threads = max(1, min(os.cpu_count() or 1, len(allowed)))   # std::thread::hardware_concurrency, here the affinity mask
    quota = _cpu_quota()   # a container's CPU bandwidth limit (cgroup): threads beyond it only take turns on the same CPU time
    cores = max(1, min(hw_threads, int(quota + 0.999))) if quota else hw_threads
    phys = min(cores, _physical_cores(allowed) or max(1, cores // 2))
    avail = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    per_key = (dim * 4 + 8) / load
    n_keys = n_keys_config
    if avail < (64 << 30):   # short box: the largest table that leaves half of the available memory alone (at least 1M keys)
        n_keys = max(1_000_000, min(n_keys_config, int(avail * 0.5 / per_key) // 1_000_000 * 1_000_000))
    t = oracle.OracleTable(int(n_keys / load), dim)
    t0 = time.perf_counter()
    placed, fill_budget_s = 0, 60.0
    while placed < n_keys:   # the same key stream and key-derived rows as the GPU table, in slices so that a slow host cannot stall the run
        c = min(10_000_000, n_keys - placed)
        got = t.populate_synth(1, placed, c, 2, threads=cores)
        assert got == c, (got, c)
        placed += c
        if time.perf_counter() - t0 > fill_budget_s and placed >= 8_000_000:
            break
    fill_s = time.perf_counter() - t0
    n_keys = placed          # (smaller than planned only if the fill ran out of its time budget: stated in `sample`)
    assert n_keys == t.size(), (n_keys, t.size())
    rng = np.random.default_rng(3)
    with np.errstate(over="ignore"):
        batches = [synth.mix64_np((rng.integers(0, n_keys, batch).astype(np.uint64) + np.uint64(1)) * np.uint64(synth._GOLDEN) + np.uint64(1)).view(np.int64)
                   for _ in range(8)]
    out, fnd = np.empty((batch, dim), np.float32), np.empty(batch, np.uint8)   # reused: a timing loop must not measure page faults
    t.find(batches[0][:4096], out=out[:4096], found=fnd[:4096])
    assert bool(fnd[:4096].all()) and np.array_equal(out[:4096], synth.rows_np(batches[0][:4096], dim, 2)), "CPU baseline returned wrong rows"
    res = {}
    # 1 thread, one per physical core, every hardware thread the process can actually run at once (affinity mask, capped by the cgroup's CPU
    # quota); under a quota also twice that, to show that threads beyond the quota only take turns
    counts = sorted({1, phys, cores} | ({min(hw_threads, 2 * cores)} if quota else {min(16, cores)}))
    for threads in counts:
        t.find(batches[0], threads=threads, out=out, found=fnd)  # warm-up (creates the pool's threads)
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(counts):
            t.find(batches[done % len(batches)], threads=threads, out=out, found=fnd)
            done += 1
        res[threads] = done * batch / (time.perf_counter() - t0)
    t.close()
    best = max(res, key=res.get)
    return {"value": res[best], "unit": "key-lookups/s", "cores": best, "kind": "port",
            "single_thread_value": res[1], "usable_hardware_threads": cores, "physical_cores": phys, "hardware_concurrency": os.cpu_count(),
            "affinity_mask_cpus": hw_threads, "cgroup_cpu_quota": quota,
            "by_threads": {str(k): v for k, v in res.items()}, "table_keys": n_keys, "table_fill_seconds": fill_s,
            "host_mem_available_gb": round(avail / 2 ** 30, 1),
            "sample": f"in-repo CPU oracle (reference snapshot has no implementation), persistent worker pool: find on a {n_keys // 1_000_000}M-key dim-{dim} "
                      f"table (load {load}; {'configs[1] size' if n_keys == n_keys_config else 'the largest that fits this host / its fill-time budget'}), uniform {batch}-key batches, "
                      f"~{budget_s / len(counts):.0f}s per thread count ({', '.join(map(str, counts))} threads; `value` = the best of them, `cores` = its thread count)"}


def kernel_window(table, batches, out, found, dev, launches=200, regions=5, warm=20, flags=None):
    """The dominant kernel alone: `regions` windows of `launches` back-to-back mee_find launches on the launch stream, HIP events
    around each window, no host sync inside -> (median, min) microseconds per launch.  Independent of --steps, so that the
    roofline object of a short driver run (--steps 20) agrees with a rocprofv3 average over hundreds of launches.
    `out` / `found`: one buffer (every launch writes the same 64 MB, which the 256 MiB Infinity Cache then absorbs) or a LIST of buffers
    the launches rotate over (independent requests with results of their own: the writes have to reach HBM)."""
    import statistics
    outs, founds = (out, found) if isinstance(out, (list, tuple)) else ([out], [found])
    nb, no = len(batches), len(outs)
    # flags: this request queue's cache policy, passed with every call (mee_find_ex); None = the library's default rule (mee_find)
    for i in range(warm):
        table.find(batches[i % nb], out=outs[i % no], found=founds[i % no], flags=flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per = []
    for r in range(regions):
        torch.cuda.synchronize(dev)
        e0.record()
        for i in range(launches):
            j = r * launches + i
            table.find(batches[j % nb], out=outs[j % no], found=founds[j % no], flags=flags)
        e1.record()
        torch.cuda.synchronize(dev)
        per.append(e0.elapsed_time(e1) * 1e3 / launches)
    return statistics.median(per), min(per)


def stream_table(table, synth, n_keys, batch, dim, dev, out, found, bpl, uniform_batches, uniform_us):
    """the same find on (i) uniform, (ii) Zipf(1.05), (iii) 90 % hit / 10 % miss key streams — median of 5
    windows of 200 launches each, plus mean probe length and the read-only GB/s column (272 B/key at dim 64)."""
    rows = {}
    streams = {"uniform": uniform_batches,
               "zipf_1.05": lookup_batches(synth, n_keys, batch, 16, "zipf", dev, seed=4)}
    mixed = lookup_batches(synth, n_keys, batch, 16, "uniform", dev, seed=5)
    g_ = torch.Generator(device=dev)
    g_.manual_seed(55)
    for b in mixed:   # 10 % of the positions ask for keys of another stream: absent
        m = torch.rand(batch, device=dev, generator=g_) < 0.10
        b[m] = synth.keys_t(99, 0, batch, dev)[m]
    streams["hit90_miss10"] = mixed
    for name, bs in streams.items():
        med, mn = (uniform_us if name == "uniform" and uniform_us else kernel_window(table, bs, out, found, dev))
        rows[name] = {"us_per_launch_median": med, "us_per_launch_min": mn, "lookups_per_s": batch / med * 1e6,
                      "algorithmic_GBps": batch * bpl / med / 1e3, "frac_of_hbm_roofline": batch * bpl / med / 1e3 / HBM_PEAK_GBS,
                      "read_only_GBps": batch * (16 + 4 * dim) / med / 1e3,
                      "mean_probe_length_buckets": table.probe_length(bs[0]),
                      # SURVEY §5 "metrics": lookups of one batch that visit 1 / 2 / 3 / 4-or-more buckets (mee_probe_histogram)
                      "probe_length_histogram": dict(zip(("1", "2", "3", "4+"), table.probe_histogram(bs[0])))}
    # the same uniform stream with the caller's per-call cache-policy hint for a stream without reuse (mee_find_ex: streaming row and bucket
    # loads, cached stores; the default keeps the loads cached because skewed streams re-read their hot rows)
    from meepoembedding_amd import _lib as _ml
    med, mn = kernel_window(table, uniform_batches, out, found, dev, flags=_ml.FIND_STREAM_ROWS | _ml.FIND_STREAM_BUCKETS | _ml.FIND_CACHED_STORES)
    rows["uniform_with_streaming_load_hint"] = {"us_per_launch_median": med, "us_per_launch_min": mn, "lookups_per_s": batch / med * 1e6,
                                                "algorithmic_GBps": batch * bpl / med / 1e3, "frac_of_hbm_roofline": batch * bpl / med / 1e3 / HBM_PEAK_GBS}
    # the same uniform stream through mee_find_unordered: launches not ordered behind each other (hipExtAnyOrderLaunch), two alternating
    # output buffers — independent requests queued on ONE stream overlap their launch latency
    o2 = [out, torch.empty_like(out)]
    f2 = [found, torch.empty_like(found)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, unordered in (("uniform_two_output_buffers_in_order", False), ("uniform_two_output_buffers_unordered_launches", True)):
        per = []
        for r in range(6):
            torch.cuda.synchronize(dev)
            e0.record()
            for i in range(200):
                table.find(uniform_batches[(r * 200 + i) % len(uniform_batches)], out=o2[i & 1], found=f2[i & 1], unordered=unordered)
            e1.record()
            torch.cuda.synchronize(dev)
            if r:
                per.append(e0.elapsed_time(e1) * 1e3 / 200)
        last = uniform_batches[(5 * 200 + 199) % len(uniform_batches)]
        assert bool(f2[1].all()) and torch.equal(o2[1][:4096], synth.rows_t(last[:4096], dim, 2)), "lookup returned wrong rows"
        per.sort()
        med = per[len(per) // 2]
        rows[name] = {"us_per_launch_median": med, "us_per_launch_min": per[0], "lookups_per_s": batch / med * 1e6,
                      "algorithmic_GBps": batch * bpl / med / 1e3, "frac_of_hbm_roofline": batch * bpl / med / 1e3 / HBM_PEAK_GBS}
    del o2, f2
    # launch-size sweep on the uniform stream: time = floor + slope * lookups (least squares) separates the per-launch latency floor
    # from the streaming rate the layout reaches
    allk = torch.cat(uniform_batches[:16])
    sizes = [1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21]
    obig = torch.empty((sizes[-1], dim), dtype=torch.float32, device=dev)
    fbig = torch.empty(sizes[-1], dtype=torch.uint8, device=dev)
    pts = []
    for sz in sizes:
        bs_ = [allk[o:o + sz] for o in range(0, allk.numel() - sz + 1, sz)][:8]
        m_, _ = kernel_window(table, bs_, obig[:sz], fbig[:sz], dev, launches=max(25, min(200, (200 << 18) // sz)), regions=3, warm=5)
        pts.append((sz, m_))
    xs, ys = np.array([p[0] for p in pts], dtype=np.float64), np.array([p[1] for p in pts])
    slope, floor = np.polyfit(xs, ys, 1)
    rows["launch_size_sweep"] = {"lookups_per_launch": [p[0] for p in pts], "us_per_launch": [p[1] for p in pts],
                                 "fit_floor_us": float(floor), "fit_us_per_262144_lookups": float(slope * 262144),
                                 "asymptotic_frac_of_hbm_roofline": float(bpl / slope / 1e3 / HBM_PEAK_GBS)}
    # north_star's batch size: 1M lookups per launch (256 MB of results per launch, two result buffers in rotation: nothing fits the Infinity
    # Cache), under the library's default cache policy and with the caller's streaming-load hint
    m1 = 1 << 20
    b1 = [allk[o:o + m1] for o in range(0, allk.numel() - m1 + 1, m1)][:4]
    o1, f1 = [obig[:m1], obig[m1:2 * m1]], [fbig[:m1], fbig[m1:2 * m1]]
    row = {"lookups_per_launch": m1, "output_buffers": 2}
    for label, fl in (("default_policy", None), ("streaming_row_load_hint", _ml.FIND_STREAM_STORES | _ml.FIND_STREAM_ROWS),
                      ("streaming_load_hint", _ml.FIND_STREAM_STORES | _ml.FIND_STREAM_ROWS | _ml.FIND_STREAM_BUCKETS)):
        med, mn = kernel_window(table, b1, o1, f1, dev, launches=50, regions=5, warm=5, flags=fl)
        row[label] = {"us_per_launch_median": med, "us_per_launch_min": mn, "lookups_per_s": m1 / med * 1e6,
                      "algorithmic_GBps": m1 * bpl / med / 1e3, "frac_of_hbm_roofline": m1 * bpl / med / 1e3 / HBM_PEAK_GBS}
    rows["north_star_batch_1M"] = row
    # the pre-exchange reductions of the sharded paths on 1M-key batches of both streams (sync-free operators, outputs preallocated): distinct keys + inverse
    # (mee_dedup_keys: what a sharded lookup sends), distinct keys + fp64-summed gradient rows (mee_dedup_sum: what a sharded backward sends)
    zk = torch.cat(streams["zipf_1.05"][:16])
    zb = [zk[o:o + m1] for o in range(0, zk.numel() - m1 + 1, m1)][:4]
    uo = torch.empty(m1, dtype=torch.int64, device=dev); co = torch.empty(m1, dtype=torch.int32, device=dev); io = torch.empty(m1, dtype=torch.int64, device=dev)
    lib_, st_ = _ml.lib(), torch.cuda.current_stream(dev).cuda_stream
    red = {"keys_per_batch": m1, "dim": dim}
    for sname, bs_ in (("uniform", b1), ("zipf_1.05", zb)):
        def dk(i):
            _ml.check(lib_.mee_dedup_keys(table._h, bs_[i % 4].data_ptr(), m1, uo.data_ptr(), io.data_ptr(), -1, st_))
        def dsum(i):
            _ml.check(lib_.mee_dedup_sum(table._h, bs_[i % 4].data_ptr(), obig[:m1].data_ptr(), m1, uo.data_ptr(), obig[m1:2 * m1].data_ptr(), co.data_ptr(), io.data_ptr(), -1, st_))
        r_ = {}
        for label, fn in (("dedup_keys_us", dk), ("dedup_sum_us", dsum)):
            for i in range(6):
                fn(i)
            per = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                e0.record()
                for i in range(20):
                    fn(i)
                e1.record()
                torch.cuda.synchronize(dev)
                per.append(e0.elapsed_time(e1) * 1e3 / 20)
            r_[label] = sorted(per)[1]
        nu = int((co > 0).sum())   # (of the last batch summed)
        assert int(co.sum()) == m1 and bool((uo[io] == bs_[19 % 4]).all()), "dedup_sum: counts / inverse of the last batch"
        r_["distinct_keys"] = nu
        r_["dedup_sum_algorithmic_GBps"] = (m1 * (16 + 4 * dim) + nu * (12 + 4 * dim)) / r_["dedup_sum_us"] / 1e3
        red[sname] = r_
    assert table.status() == 0
    rows["pre_exchange_reductions_1M"] = red
    del obig, fbig, allk, zk
    return rows


def tier_1b_point(synth, dim, dev, batch, log, hot_keys=800_000_000, cold_keys=200_000_000, load=0.85):
    """The literal 1B-key dim-64 table on ONE GPU (BASELINE metric; it does not fit 288 GB of HBM at any load factor): 800M keys in an HBM
    table backed by 200M keys whose rows live in pinned host DRAM (the hot/cold tier of configs[4]); uniform lookups over all 1B keys
    (20 % of them cold: PCIe-bound) and over the hot keys only."""
    from meepoembedding_amd import LookupTable, _lib
    from meepoembedding_amd.tiered import TieredLookupTable
    chunk = 1 << 20
    t0 = time.time()
    hot = LookupTable(int(hot_keys / load), dim, device=dev, max_batch=chunk)
    cold = LookupTable(int(cold_keys / load), dim, device=dev, max_batch=chunk, value_memory=_lib.MEM_HOST_PINNED)
    for s_ in range(0, hot_keys, chunk):
        k = synth.keys_t(1, s_, min(chunk, hot_keys - s_), dev)
        hot.insert(k, synth.rows_t(k, dim, 2))
    for s_ in range(0, cold_keys, chunk):
        k = synth.keys_t(1, hot_keys + s_, min(chunk, cold_keys - s_), dev)
        cold.insert(k, synth.rows_t(k, dim, 2))
    torch.cuda.synchronize(dev)
    log(f"tier point: {hot_keys + cold_keys} keys placed in {time.time() - t0:.1f}s (hot {hot.table_bytes / 1e9:.0f} GB HBM, cold rows {cold_keys * dim * 4 / 1e9:.0f} GB pinned host)")
    tiered = TieredLookupTable(hot, cold, hot_key_limit=hot_keys)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    res = {"keys": hot_keys + cold_keys, "hot_keys_in_hbm": hot_keys, "cold_keys_in_pinned_host": cold_keys, "load_factor": load,
           "hbm_gb": round((hot.table_bytes + cold_keys / load * 8) / 1e9, 1)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, span in (("uniform_over_all_keys", hot_keys + cold_keys), ("uniform_over_hot_keys", hot_keys)):
        bs = [synth.mix64_t((torch.randint(0, span, (batch,), device=dev, generator=g) + 1) * synth._s64(synth._GOLDEN) + synth._s64(1)) for _ in range(8)]
        o_, f_ = tiered.find(bs[0])
        assert bool(f_.all()) and torch.equal(o_[:2048], synth.rows_t(bs[0][:2048], dim, 2)), "tier lookup returned wrong rows"
        per = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            e0.record()
            for i in range(20):
                tiered.find(bs[i % 8])
            e1.record()
            torch.cuda.synchronize(dev)
            per.append(e0.elapsed_time(e1) * 1e3 / 20)
        us = sorted(per)[1]
        res[name] = {"us_per_find": us, "lookups_per_s": batch / us * 1e6, "cold_fraction": max(0.0, 1.0 - hot_keys / span),
                     "cold_rows_over_pcie_GBps": batch * max(0.0, 1.0 - hot_keys / span) * dim * 4 / us / 1e3}
    hot.close(); cold.close()
    return res


def _host_ram_available_gb() -> float:
    """what this process may still take of the host's RAM: MemAvailable, cut by the cgroup's limit when there is one"""
    avail = 0.0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = float(ln.split()[1]) * 1024 / 1e9
    except OSError:
        pass
    for mx, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                    ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(mx).read().strip()
            if m != "max" and int(m) < (1 << 60):
                left = (int(m) - int(open(cur).read().strip())) / 1e9
                avail = min(avail, left) if avail else left
        except (OSError, ValueError):
            pass
    return avail


def tier_share_point(synth, dim, dev, batch, log, keys=1_250_000_000, hot_keys=800_000_000, train_cold_cap=220_000_000, load=0.85, rounds=8):
    """configs[4]'s per-GPU share on ONE GPU: 10B keys over 8 GPUs = 1.25B keys, dim 64: 800M of them in an HBM table, 450M with their rows in
    pinned host DRAM (115 GB; their key plane stays in HBM), both with hit counters.  A Zipf(1.05) stream whose popular ranks are scattered
    over ALL key indices (a third of the popular keys start cold); observe -> rebalance() rounds until the cold share of the lookups stops
    falling; lookups/s, cold share and PCIe rate before and after.  Then one find + sparse-Adagrad step on a training pair — rows AND
    accumulators: 520 B per slot, so the pair that fits the same HBM holds 380M hot keys, and as many cold keys as the host RAM left."""
    from meepoembedding_amd import LookupTable, OPT_ADAGRAD, _lib
    from meepoembedding_amd.tiered import TieredLookupTable
    chunk = 1 << 20
    cold_keys = keys - hot_keys
    res = {"keys_wanted": keys, "hot_keys_in_hbm": hot_keys, "load_factor": load, "batch": batch, "dim": dim}
    ram = _host_ram_available_gb()
    need = cold_keys * dim * 4 / 1e9
    res["host_ram_available_gb"] = round(ram, 1)
    res["cold_rows_wanted_gb"] = round(need, 1)
    if ram and need > ram * 0.5:   # never pin more than half of what the box may take: pinned pages cannot be reclaimed, and a box that runs out of memory dies
        cold_keys = max(chunk, int(ram * 0.5 * 1e9 / (dim * 4)) // chunk * chunk)
        res["scaled_down"] = f"the box offers {ram:.0f} GB of host RAM: {cold_keys} cold keys ({cold_keys * dim * 4 / 1e9:.0f} GB pinned) instead of {keys - hot_keys}"
        log("tier share: " + res["scaled_down"])
    keys = hot_keys + cold_keys
    res["keys"] = keys
    res["cold_keys_in_pinned_host"] = cold_keys
    t0 = time.time()
    hot = LookupTable(int(hot_keys / load), dim, device=dev, max_batch=4 * chunk, track_hits=True)
    cold = LookupTable(int(cold_keys / load), dim, device=dev, max_batch=4 * chunk, value_memory=_lib.MEM_HOST_PINNED, track_hits=True)
    for tab, lo, hi_ in ((hot, 0, hot_keys), (cold, hot_keys, keys)):   # key index i: the first hot_keys indices land in HBM, the rest in the cold tier
        for s_ in range(lo, hi_, chunk):
            k = synth.keys_t(1, s_, min(chunk, hi_ - s_), dev)
            tab.insert(k, synth.rows_t(k, dim, 2))
            if (s_ - lo) // chunk % 128 == 127:
                torch.cuda.synchronize(dev)
                log(f"tier share: {s_ + chunk - lo} of {hi_ - lo} keys of the {'HBM' if tab is hot else 'host-DRAM'} tier placed ({time.time() - t0:.0f}s)")
    torch.cuda.synchronize(dev)
    res["populate_s"] = round(time.time() - t0, 1)
    res["hbm_gb"] = round((hot.table_bytes + cold_keys / load * 8 + (hot_keys + cold_keys) / load * 4) / 1e9, 1)
    log(f"tier share: {keys} keys placed in {res['populate_s']}s (hot {hot.table_bytes / 1e9:.0f} GB HBM, cold rows {cold_keys * dim * 4 / 1e9:.0f} GB pinned host)")
    t = TieredLookupTable(hot, cold, hot_key_limit=hot_keys, sample_every=4, promote_threshold=2)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    mult = 2_654_435_761   # odd: popular ranks are scattered over all key indices (where two ranks collide they merge)

    def zipf_batch(n_keys):
        al = 1.05
        u = torch.rand(batch, device=dev, generator=g, dtype=torch.float64)
        hi = float(n_keys) ** (1 - al)
        r = ((1 + u * (hi - 1)) ** (1 / (1 - al))).floor().to(torch.int64).clamp_(1, n_keys) - 1
        idx = (r * mult + 12345) % n_keys
        return synth.mix64_t((idx + 1) * synth._s64(synth._GOLDEN) + synth._s64(1))

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(tt, hot_t, n_keys, n=20):
        bs = [zipf_batch(n_keys) for _ in range(n)]
        o_, f_ = tt.find(bs[0])
        assert bool(f_.all()) and torch.equal(o_[:2048], synth.rows_t(bs[0][:2048], dim, 2)), "tier lookup returned wrong rows"
        torch.cuda.synchronize(dev)
        e0.record()
        for b in bs:
            tt.find(b)
        e1.record()
        torch.cuda.synchronize(dev)
        cs = sum(1.0 - float(hot_t.find(b)[1].float().mean()) for b in bs[:4]) / 4
        us = e0.elapsed_time(e1) * 1e3 / n
        return {"us_per_find": us, "lookups_per_s": batch / us * 1e6, "cold_fraction": cs, "cold_rows_over_pcie_GBps": batch * cs * dim * 4 / us / 1e3}

    res["before_rebalance"] = measure(t, hot, keys)
    log(f"tier share: before rebalance {res['before_rebalance']}")
    hist = []
    for rnd in range(1, rounds + 1):
        for _ in range(24):
            t.find(zipf_batch(keys))          # observation window (every 4th hot lookup sampled, all cold hits counted)
        t1 = time.time()
        p_, d_ = t.rebalance(max_moves=4 * chunk)
        torch.cuda.synchronize(dev)
        row = measure(t, hot, keys)
        row.update({"round": rnd, "promoted": int(p_), "demoted": int(d_), "rebalance_ms": round((time.time() - t1) * 1e3, 1)})
        hist.append(row)
        log(f"tier share: round {rnd}: {row}")
        if len(hist) >= 3 and hist[-2]["cold_fraction"] - row["cold_fraction"] < 0.02 * max(row["cold_fraction"], 1e-9):
            break   # plateau: less than 2 % (relative) off the cold share in a round
    res["rebalance_rounds"] = hist
    res["after_rebalance"] = {k: v for k, v in hist[-1].items() if k not in ("round", "promoted", "demoted", "rebalance_ms")}
    assert hot.status() == 0 and cold.status() == 0
    hot.close(); cold.close()
    del t, hot, cold
    torch.cuda.empty_cache()
    # ---- the training pair: values + Adagrad accumulators ----
    th_keys = min(380_000_000, hot_keys * 19 // 40)
    ram = _host_ram_available_gb()
    tc_keys = min(keys - th_keys, max(chunk, int(ram * 0.5 * 1e9 / (dim * 8)) // chunk * chunk)) if ram else 120_000_000
    if train_cold_cap:   # (default: no more pinned host memory than the lookup tables above took — a box shares its host with seven others)
        tc_keys = min(tc_keys, train_cold_cap)
    t0 = time.time()
    hot = LookupTable(int(th_keys / load), dim, device=dev, max_batch=4 * chunk, optimizer=OPT_ADAGRAD, track_hits=True)
    cold = LookupTable(int(tc_keys / load), dim, device=dev, max_batch=4 * chunk, optimizer=OPT_ADAGRAD, value_memory=_lib.MEM_HOST_PINNED, track_hits=True)
    tk = th_keys + tc_keys
    for tab, lo, hi_ in ((hot, 0, th_keys), (cold, th_keys, tk)):
        for s_ in range(lo, hi_, chunk):
            k = synth.keys_t(1, s_, min(chunk, hi_ - s_), dev)
            tab.insert(k, synth.rows_t(k, dim, 2))
            if (s_ - lo) // chunk % 128 == 127:
                torch.cuda.synchronize(dev)
                log(f"tier share: {s_ + chunk - lo} of {hi_ - lo} keys of the {'HBM' if tab is hot else 'host-DRAM'} tier placed ({time.time() - t0:.0f}s)")
    torch.cuda.synchronize(dev)
    tt = TieredLookupTable(hot, cold, hot_key_limit=th_keys)
    tr = {"hot_keys_in_hbm": th_keys, "cold_keys_in_pinned_host": tc_keys, "bytes_per_slot": 8 + 2 * dim * 4, "populate_s": round(time.time() - t0, 1),
          "cold_rows_and_accumulators_gb": round(tc_keys * dim * 8 / 1e9, 1),
          "note": ("the whole share" if th_keys + tc_keys >= keys else f"rows + accumulators of the whole share need {(keys - th_keys) * dim * 8 / 1e9:.0f} GB of host RAM; this is the pair that fits")}
    grads = torch.randn(batch, dim, device=dev) * 0.01
    for name, sk in (("zipf_scattered", True), ("zipf_hot_first", False)):
        if sk:
            bs = [zipf_batch(tk) for _ in range(8)]
        else:
            bs = lookup_batches(synth, tk, batch, 8, "zipf", dev, seed=5)   # popular ranks = the first key indices = the HBM tier
        cs = sum(1.0 - float(hot.find(b)[1].float().mean()) for b in bs[:4]) / 4
        cold_distinct = sum(int(torch.unique(b[hot.find(b)[1] == 0]).numel()) for b in bs[:4]) / 4
        per = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            e0.record()
            for i in range(10):
                tt.find(bs[i % 8])
                tt.apply_adagrad(bs[i % 8], grads, lr=0.01)
            e1.record()
            torch.cuda.synchronize(dev)
            per.append(e0.elapsed_time(e1) * 1e3 / 10)
        us = sorted(per)[1]
        tr[name] = {"us_per_find_plus_adagrad_step": us, "keys_per_s": batch / us * 1e6, "cold_fraction": cs,
                    "cold_distinct_keys_per_batch": cold_distinct,
                    # the find reads a row per cold occurrence; the update reads and writes row + accumulator once per distinct cold key
                    "cold_bytes_over_pcie_GBps": (batch * cs + 4 * cold_distinct) * dim * 4 / us / 1e3}
        log(f"tier share: training pair, {name}: {tr[name]}")
    # ---- the training loop with the placement policy IN it (round 5): rebalance_every = 24 optimizer steps.  The scattered stream starts with half of its
    # lookups cold; every round = 24 find + Adagrad steps (the observation window), the rebalance behind the 24th (keys move WITH their accumulators), then
    # the step is timed again (policy off while it is timed)
    tp = TieredLookupTable(hot, cold, hot_key_limit=th_keys, sample_every=4, promote_threshold=2, rebalance_every=24, rebalance_max_moves=4 * chunk)
    series = []

    def timed_steps(bs_):   # (policy AND observation paused: the timed batches must not vote for their own keys)
        tp.rebalance_every, keep = 0, tp.rebalance_every
        tp.policy, keep_policy = False, tp.policy
        per = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            e0.record()
            for i in range(10):
                tp.find(bs_[i % len(bs_)])
                tp.apply_adagrad(bs_[i % len(bs_)], grads, lr=0.01)
            e1.record()
            torch.cuda.synchronize(dev)
            per.append(e0.elapsed_time(e1) * 1e3 / 10)
        tp.rebalance_every, tp.policy = keep, keep_policy
        return sorted(per)[1]

    for rnd in range(0, rounds + 1):
        probe_bs = [zipf_batch(tk) for _ in range(8)]   # fresh draws of the stream every round: their once-seen tail keys have never been observed
        if rnd:
            t1 = time.time()
            for _ in range(24):
                b_ = zipf_batch(tk)
                tp.find(b_)
                tp.apply_adagrad(b_, grads, lr=0.01)
            torch.cuda.synchronize(dev)
            loop_ms = (time.time() - t1) * 1e3
        cs = sum(1.0 - float(hot.find(b)[1].float().mean()) for b in probe_bs[:4]) / 4
        row = {"round": rnd, "us_per_find_plus_adagrad_step": timed_steps(probe_bs), "cold_fraction": cs}
        if rnd:
            row.update({"promoted": tp.rebalance_log[-1][1], "demoted": tp.rebalance_log[-1][2], "ms_for_24_steps_and_the_rebalance": round(loop_ms, 1)})
        series.append(row)
        log(f"tier share: training loop with rebalance_every=24, round {rnd}: {row}")
        if len(series) >= 4 and series[-2]["cold_fraction"] - cs < 0.02 * max(cs, 1e-9):
            break
    tr["training_loop_with_policy"] = {"rebalance_every_steps": 24, "rounds": series}
    assert hot.status() == 0 and cold.status() == 0
    res["train_pair"] = tr
    hot.close(); cold.close()
    return res


def two_stream_extra(table, batches, dim, dev, bpl, launches=400):
    """The same find launches issued round-robin on TWO caller streams (two independent request queues, own output
    buffers): consecutive launches may overlap each other's latency floor.  Informational — the headline keeps every
    step on one stream, which is also what `roofline` prices."""
    batch = batches[0].numel()
    main_s = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    outs = [torch.empty((batch, dim), dtype=torch.float32, device=dev) for _ in range(2)]
    founds = [torch.empty(batch, dtype=torch.uint8, device=dev) for _ in range(2)]

    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % 2]):
                table.find(batches[i % len(batches)], out=outs[i % 2], found=founds[i % 2])

    for s_ in streams:
        s_.wait_stream(main_s)
    run(20)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(launches)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / launches
    res = {"lookups_per_s": batch / dt, "us_per_batch": dt * 1e6, "frac_of_hbm_roofline": batch * bpl / dt / 1e9 / HBM_PEAK_GBS}
    # four queued requests served by ONE launch (mee_find_many): the launch floor is paid once per four batches, on one stream
    outs4 = [torch.empty((batch, dim), dtype=torch.float32, device=dev) for _ in range(4)]
    founds4 = [torch.empty(batch, dtype=torch.uint8, device=dev) for _ in range(4)]

    def run4(k):
        for i in range(k):
            table.find_many([(batches[(4 * i + j) % len(batches)], outs4[j], founds4[j]) for j in range(4)])

    run4(5)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run4(launches // 4)
    torch.cuda.synchronize(dev)
    dt4 = (time.perf_counter() - t0) / (launches // 4) / 4
    res["four_requests_per_launch"] = {"lookups_per_s": batch / dt4, "us_per_batch": dt4 * 1e6, "frac_of_hbm_roofline": batch * bpl / dt4 / 1e9 / HBM_PEAK_GBS}
    return res


def p2p_selftest(ctrl, log, timeout=120, script="p2p_selftest.py", port_offset=17, what="p2p") -> bool:
    """tools/<script> in one child per rank (own gloo group, same GPUs); True only if every child exits 0."""
    import subprocess
    # the children rendezvous among themselves: their rank 0 hosts a store of its own (torchrun's agent store is not theirs)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29531")) + port_offset)
    t0 = time.time()
    err = b""
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], env=env, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        rc, err = p.returncode, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, err = -9, e.stderr or b""
    if rc != 0:
        print(f"[bench] {what} self-test child of rank {os.environ.get('RANK', '0')} rc={rc}: {err.decode(errors='replace')[-800:]}", file=sys.stderr, flush=True)
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=ctrl)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    log(f"{what} self-test: rc={rc} on this rank, {'passed on all ranks' if int(ok.item()) else 'FAILED somewhere -> not used'} ({time.time() - t0:.1f}s)")
    return bool(int(ok.item()))


def train_streams(table, synth, n_keys, batch, dim, dev, out, found, bpl, steps=60, regions=3, opt="adagrad"):
    """the find + sparse-Adagrad step on a uniform and a Zipf(1.05) key stream, bytes by the 264*B + 1032*U rule
    with U measured; median of `regions` HIP-event windows of `steps` steps each (after the timed region: not part of `value`)."""
    res = {}
    slots = torch.empty(batch, dtype=torch.int64, device=dev)
    grads = [torch.randn(batch, dim, device=dev) * 0.01 for _ in range(4)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, dist_name in (("uniform", "uniform"), ("zipf_1.05", "zipf")):
        bs_ = lookup_batches(synth, n_keys, batch, 8, dist_name, dev, seed=11)
        uniq = sum(int(torch.unique(b_).numel()) for b_ in bs_) / len(bs_)

        step_no = [0]

        def apply(i, **kw):
            if opt == "adagrad":
                table.apply_adagrad(bs_[i % 8], grads[i % 4], lr=0.01, eps=1e-10, **kw)
            else:
                step_no[0] += 1
                table.apply_adam(bs_[i % 8], grads[i % 4], lr=0.001, step=step_no[0], **kw)

        def step(i):   # the training forward (its launch also partitions the batch for the backward), then the backward on the located slots
            table.find_located(bs_[i % 8], out=out, found=found, slots=slots, prepare_apply=fused_forward[0])
            apply(i, slots=slots)

        def apply_only(i):
            apply(i)

        row = {"unique_keys_per_batch": uniq}
        fused_forward = [True]
        per_u = 8 + (16 if opt == "adagrad" else 24) * dim   # table key + w, state read and written (1032 B Adagrad / 1544 B Adam at dim 64)
        sfx = "" if opt == "adagrad" else "_adam"
        for label, fn, nbytes in (("step" + sfx, step, (bpl + 8 + 4 * dim) * batch + per_u * uniq),
                                  ("step_separate_forward" + sfx, step, (bpl + 8 + 4 * dim) * batch + per_u * uniq),
                                  ("apply_alone" + sfx, apply_only, (8 + 4 * dim) * batch + per_u * uniq)):
            fused_forward[0] = label.startswith("step") and "separate" not in label   # "step": mee_find_located_prepare + mee_apply_*_located; "step_separate_forward": mee_find_located + the apply
            ts = []
            for _ in range(regions):
                for i in range(5):
                    fn(i)
                torch.cuda.synchronize(dev)
                e0.record()
                for i in range(steps):
                    fn(i)
                e1.record()
                torch.cuda.synchronize(dev)
                ts.append(e0.elapsed_time(e1) * 1e3 / steps)
            us = sorted(ts)[len(ts) // 2]
            row[label] = {"us": us, "keys_per_s": batch / us * 1e6, "algorithmic_bytes_per_key": nbytes / batch,
                          "frac_of_hbm_roofline": nbytes / us / 1e3 / HBM_PEAK_GBS}
        res[name] = row
    return res


def configs2_rows(find_table, synth, n_keys, dim, dev, chunk, batch, bpl):
    """configs[2] on the same box, carried by the DEFAULT line: the find table is replaced by one with an Adagrad plane (69 GB at 100M keys),
    then SURVEY 8d config 3's table is measured — find_located + sparse-Adagrad apply on the located slots per step, and the apply alone,
    on a uniform and a Zipf(1.05) key stream, bytes by 528*B + 264*B + 1032*U with U measured."""
    from meepoembedding_amd import OPT_ADAGRAD, LookupTable
    cap = find_table.capacity
    find_table.close()   # 35 GB back before the 69 GB table is made
    torch.cuda.empty_cache()
    t = LookupTable(cap, dim, device=dev, max_batch=max(chunk, 2 * batch), optimizer=OPT_ADAGRAD)
    populate(t, synth, n_keys, dim, dev, chunk)
    out = torch.empty((batch, dim), dtype=torch.float32, device=dev)
    found = torch.empty(batch, dtype=torch.uint8, device=dev)
    rows = train_streams(t, synth, n_keys, batch, dim, dev, out, found, bpl)
    rows["workload"] = (f"configs[2]: 1xMI355X, {n_keys // 1_000_000}M keys, dim {dim}, forward gather (mee_find_located) + sparse-Adagrad backward scatter-update on the "
                        f"located slots, {batch}-key batches, N(0, 1e-2) grads, lr 0.01; {t.table_bytes / 1e9:.1f} GB table; median of 3 HIP-event windows of 60 steps; "
                        f"the *_adam rows: the same with sparse Adam (lazy: untouched rows' moments not decayed) on a table with two state planes")
    t.close()
    torch.cuda.empty_cache()
    try:   # north_star "Adagrad/Adam": the same rows with sparse Adam (103 GB table: rows + two moment planes)
        from meepoembedding_amd import OPT_ADAM
        ta = LookupTable(cap, dim, device=dev, max_batch=max(chunk, 2 * batch), optimizer=OPT_ADAM)
        populate(ta, synth, n_keys, dim, dev, chunk)
        adam = train_streams(ta, synth, n_keys, batch, dim, dev, out, found, bpl, opt="adam")
        for name_, row_ in adam.items():
            rows[name_].update({k_: v_ for k_, v_ in row_.items() if k_.endswith("_adam")})
        rows["adam_table_gb"] = round(ta.table_bytes / 1e9, 1)
        ta.close()
    except Exception as e:  # noqa: BLE001
        rows["adam_error"] = repr(e)
    return rows


def visible_gpu_count() -> int:
    """GPUs this process could use, WITHOUT touching the HIP runtime (the launcher forks its ranks right after: nothing that may open the
    device belongs in front of that): KFD topology nodes with SIMDs, cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES if they are set."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for ln in open(f):
                if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                    n += 1
        except OSError:
            pass
    if n == 0:
        n = len(glob.glob("/dev/dri/renderD*"))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves (one per GPU, torch.distributed.run on
    127.0.0.1), relay rank 0's single JSON line to our stdout and return the launcher's exit code (non-zero if any rank failed).
    Runs BEFORE this process has made any HIP call: the ranks are fresh children, nothing is re-executed over a GPU context."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL and the peer-mapped transport need it on this driver
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)   # stderr passes through
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        print("[bench] the ranks exited 0 but printed no result line", file=sys.stderr)
        return 1
    return p.returncode


def link_traffic(router, batches, carrier, cap_padded, dim, world, ctrl, payload_dim=0):
    """Bytes each ordered GPU pair (a -> b) carries per step, from the owner counts of a representative batch (all-gathered):
    keys of a's lookups that b owns travel a -> b, the rows b serves for them travel b -> a.  SURVEY 8e: the fully connected xGMI mesh
    gives every pair its own link, so the busiest link-direction bounds the step."""
    _, counts, _ = router.partition(batches[0])
    mine = counts.to(torch.int64).to(ctrl)
    allc = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    c = torch.stack(allc).cpu().numpy().astype(np.float64)   # c[a][b]: keys rank a sends to owner b
    back = 4 * dim + 1                                        # a row and its found byte
    if carrier == "native" and cap_padded:
        out_b = np.full_like(c, 8.0 * cap_padded)             # constant-size padded segments
        back_b = np.full_like(c, float(back) * cap_padded)
    elif carrier == "p2p":
        out_b, back_b = c * 16.0, c * back                    # key + destination index pushed, rows stored straight into the requester
    else:
        out_b, back_b = c * 8.0 + 8.0, c * back               # counts word + keys out, rows + found back
    out_b = out_b + c * 4.0 * payload_dim                     # train mode: the gradient rows follow the keys
    per_link = out_b + back_b.T                               # a -> b carries a's keys to b and the rows a serves for b
    np.fill_diagonal(per_link, 0.0)
    return float(per_link.max()), float(per_link.sum())


XGMI_LINK_GBS_PER_DIR = 76.8   # 153.6 GB/s per link bidirectional (nominal; re-measure with a peer-copy microbenchmark)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--keys", type=int, default=None, help="keys per GPU (default 100M at N=1, 125M sharded)")
    ap.add_argument("--batch", type=int, default=None, help="lookups per rank per step (default 256K at N=1, 1M sharded)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--load", type=float, default=0.75)
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="append an `also` object: the same find launches on two caller streams and four requests per launch")
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)  # accepted for older command lines
    ap.add_argument("--mode", choices=["find", "train"], default="find",
                    help="find = configs[1] (the driver's metric); train = configs[2]: find + sparse-Adagrad apply per step (sharded: gradients travel to the owners too)")
    ap.add_argument("--pipeline", type=int, default=1, help="sharded only: steps in flight on separate HIP streams (1 = off)")
    ap.add_argument("--transport", choices=["auto", "rccl", "native", "p2p"], default="auto",
                    help="sharded only: rccl = torch.distributed all-to-all exchange; native = the same exchange behind the C-ABI (mee_sharded_*: "
                         "grouped ncclSend/ncclRecv inside the library; exact and padded segment layouts); p2p = owners store rows into the "
                         "requester's peer-mapped buffer; auto = verify each against the torch.distributed path, time them for a few steps, keep the fastest")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 flow on a box with fewer GPUs than ranks (exchange staged through host memory)")
    ap.add_argument("--p2p-depth", type=int, default=2, help="sharded, peer-mapped transport: also try this many lookups in flight (own context + stream each) and keep it if faster; 1 = off")
    ap.add_argument("--no-selftest", action="store_true", help="sharded only: skip the child-process self-test of the peer-mapped transport")
    ap.add_argument("--dedup", action="store_true", help="sharded only: exchange only the batch's distinct keys (pays off on skewed streams)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-sharded path even at N=1 (rehearsal of the N>1 code)")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph",
                    help="N=1 find mode: how the K timed steps are issued — one hipGraph replay of K chained kernel nodes (default) or K host launch calls")
    ap.add_argument("--no-streams", action="store_true", help="skip the SURVEY 8d key-stream table (uniform / Zipf / 90-10 hit-miss) and the configs[2] rows in the JSON line")
    ap.add_argument("--out-buffers", type=int, default=6,
                    help="N=1 find mode: the timed lookups rotate over this many dense output buffers (6 x 64 MB > the 256 MiB Infinity Cache: every "
                         "result has to reach HBM, as the results of independent requests do); 1 = every step overwrites one buffer")
    ap.add_argument("--no-configs2", action="store_true", help="N=1 find mode: skip the configs[2] (find + sparse Adagrad) rows that the default line carries")
    ap.add_argument("--tier-1b", action="store_true",
                    help="N=1 only: after the headline, also measure the LITERAL 1B-key dim-64 table on this one GPU through the hot/cold tier "
                         "(800M keys in HBM + 200M keys with rows in pinned host DRAM; needs ~270 GB of HBM and ~55 GB of pinned host memory); result under `also`")
    ap.add_argument("--tier-share", action="store_true",
                    help="N=1: also run configs[4]'s per-GPU share (1.25B keys: 800M in HBM + 450M rows in pinned host DRAM, Zipf, rebalance rounds; "
                         "needs ~260 GB of HBM and ~115 GB of host RAM — scaled to what the box has) -> also.tier_share_one_gpu")
    ap.add_argument("--tier-share-keys", type=str, default="1250000000,800000000", help="--tier-share: total keys, keys in HBM[, cap on the cold keys of the training pair] (smaller = a rehearsal)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "LOCAL_RANK" not in os.environ:
        # no launcher around us: become the launcher.  Nothing in this process has touched the GPU yet (the GPUs are counted from sysfs).
        have = visible_gpu_count()
        if args.backend == "nccl" and 0 < have < args.gpus:   # (0 = sysfs told nothing: let the ranks find out)
            raise SystemExit(f"--gpus {args.gpus} over RCCL needs {args.gpus} GPUs, {have} visible "
                             f"(--backend gloo rehearses the N>1 flow with ranks sharing the GPUs there are)")
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner to stdout when the process
    # group is created) are pointed at stderr for the whole run, and the result is written to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")

    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    ctrl = torch.device("cpu") if args.backend == "gloo" else dev   # where the small control tensors of collectives live
    if world > 1 or args.force_sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # rank 0 (re)builds the native library if it is stale; nobody loads it before that is done
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if dist.is_initialized():
        dist.barrier()
    from meepoembedding_amd import LookupTable, Router, hash_batch, synth
    from meepoembedding_amd.sharded import ShardedLookupTable

    t_start = time.time()

    def log(msg):   # --verbose: every phase with the wall time since the start of the process (populate, self-tests, probes, timed region ...)
        if args.verbose and rank == 0:
            print(f"[bench +{time.time() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    sharded = world > 1 or args.force_sharded
    keys_per_gpu = args.keys or (125_000_000 if sharded else 100_000_000)
    batch = args.batch or ((1 << 20) if sharded else (1 << 18))
    n_keys = keys_per_gpu * world
    dim = args.dim
    chunk = 1 << 20
    cap = int(keys_per_gpu / args.load * (1.02 if sharded else 1.0))  # shard sizes fluctuate a little around n/world
    train = args.mode == "train"
    from meepoembedding_amd import OPT_ADAGRAD, OPT_NONE
    table = LookupTable(cap, dim, device=dev, max_batch=max(chunk, batch * 2), optimizer=OPT_ADAGRAD if train else OPT_NONE)
    log(f"table: {table.capacity} slots, {table.table_bytes / 1e9:.1f} GB")
    populate(table, synth, n_keys, dim, dev, chunk, owner_rank=rank, world=world, hash_batch=hash_batch, log=log)
    local_size = table.size()
    assert table.status() == 0, "table full / reserved key during populate"
    n_batches = 16 if sharded else 64
    batches = lookup_batches(synth, n_keys, batch, n_batches, args.dist, dev, seed=3 + rank)
    out = torch.empty((batch, dim), dtype=torch.float32, device=dev)
    found = torch.empty(batch, dtype=torch.uint8, device=dev)
    # N=1 find mode: the timed steps rotate over several output buffers (results of independent requests; more bytes than the Infinity Cache holds)
    n_out = max(1, args.out_buffers) if (not sharded and not train) else 1
    outs = [out] + [torch.empty_like(out) for _ in range(n_out - 1)]
    founds = [found] + [torch.empty_like(found) for _ in range(n_out - 1)]
    store_hint = None   # the caller's cache-policy hint for the rotating case, chosen by a short probe below
    find_flags = None   # the cache policy this request queue passes with every call (mee_find_ex); None = the library's default rule

    if sharded:
        # each in-flight step owns a stream and a Router (partition workspace); the local find is workspace-free, so
        # lookups of different steps may overlap: the all-to-all of one step runs beside the gather of the next
        depth = max(1, args.pipeline)
        shs = [ShardedLookupTable(table, Router(world, batch, device=dev)) for _ in range(depth)]
        streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(depth - 1)]

        def step_rccl(i):
            if depth == 1:
                return shs[0].find(batches[i % n_batches], dedup=args.dedup)
            with torch.cuda.stream(streams[i % depth]):
                return shs[i % depth].find(batches[i % n_batches], dedup=args.dedup)

        def timed(fn, k=6):
            for i in range(3):
                fn(i)
            dist.barrier(); torch.cuda.synchronize(dev)
            t_ = time.perf_counter()
            for i in range(k):
                fn(i)
            torch.cuda.synchronize(dev)
            tt_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=ctrl)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            return float(tt_.item()) / k

        step, transport = step_rccl, "rccl all-to-all (torch.distributed)"
        carrier = "rccl"     # which transport carries the timed steps: "rccl" (torch.distributed), "native" (behind the C-ABI) or "p2p"
        peer = None
        peer_ctxs = []       # every peer-mapped context the timed steps use (several when lookups are kept in flight)
        t_best = None        # seconds per step of `step`, once something has been timed against it
        native = None        # the RcclShardedTable that carries `step`, if any
        verified_native = [] # (label, context) of every native layout that matched the torch.distributed path bit for bit in the probe
        p2p_verified = False # the peer-mapped context matched it too
        native_tables = []   # every native context created by the probe: closed (communicators destroyed) before the process group goes
        # ---- the exchange behind the C-ABI (mee_sharded_*: grouped ncclSend/ncclRecv inside the library), both layouts ----
        # (the gloo rehearsal on one GPU can take this path too when MEE_RCCL_LIB names the test suite's shared-memory stand-in for RCCL)
        native_ok = args.transport in ("auto", "native") and depth == 1 and (args.backend == "nccl" or os.environ.get("MEE_RCCL_LIB"))
        if native_ok and not args.no_selftest:
            native_ok = p2p_selftest(ctrl, log, script="rccl_selftest.py", port_offset=23, what="native rccl")
        if native_ok:
            from meepoembedding_amd.sharded import RcclShardedTable
            t_best = timed(step_rccl)
            log(f"transport probe: torch.distributed all-to-all {t_best * 1e3:.3f} ms/step")
            for label, slack in (("exact segments, one host sync per lookup", 0.0), ("padded segments, no host sync", 1.02)):   # +2 % + 1024 positions per segment: 10 sigma of a uniform batch at 131K..1M keys per rank
                try:
                    nt = RcclShardedTable(table, batch, pad_slack=slack, dedup=args.dedup)   # collective (ncclCommInitRank): proven by the self-test
                except Exception as e:  # noqa: BLE001
                    log(f"native rccl ({label}) unavailable: {e}")
                    break
                native_tables.append(nt)

                def step_nat(i, nt=nt):
                    return nt.find(batches[i % n_batches])

                o_a, f_a = step_rccl(0)
                o_b, f_b = step_nat(0)
                okn = torch.tensor([int(torch.equal(o_a, o_b) and torch.equal(f_a, f_b) and nt.status() == 0)], device=ctrl)
                dist.all_reduce(okn, op=dist.ReduceOp.MIN)
                if int(okn.item()) != 1:
                    log(f"native rccl ({label}): differs from the torch.distributed path or a segment overflowed: not used")
                    continue
                verified_native.append((label, nt))
                t_n = timed(step_nat)
                log(f"transport probe: native rccl ({label}) {t_n * 1e3:.3f} ms/step")
                if t_n < t_best or (args.transport == "native" and native is None):
                    step, transport, t_best, native, carrier = step_nat, f"RCCL grouped send/recv behind the C-ABI ({label})", t_n, nt, "native"
        p2p_ok = args.transport in ("auto", "p2p") and depth == 1
        if p2p_ok and not args.no_selftest:
            # the peer-mapped path stores into other GPUs' memory from hand-written kernels: prove it on THIS topology in
            # throw-away child processes first, so that a fault or hang there costs the fallback, not the run
            p2p_ok = p2p_selftest(ctrl, log)
        if p2p_ok:
            from meepoembedding_amd.p2p import PeerShardedFind
            try:
                peer = PeerShardedFind(table, Router(world, batch, device=dev), max_batch=batch, payload=train)
            except Exception as e:  # collective failure: every rank lands here together
                log(f"p2p transport unavailable: {e}")
            if peer is not None:
                def step_p2p(i):
                    return peer.find(batches[i % n_batches], check_overflow=False, dedup=args.dedup)

                o_a, f_a = step_rccl(0)
                o_b, f_b = step_p2p(0)
                ok_probe = int(torch.equal(o_a, o_b) and torch.equal(f_a, f_b))
                try:   # an inbox overflow (skewed keys without --dedup) raises only on the ranks whose push overflowed:
                    peer.check()   # every rank must reach the all-reduce below, or the others hang in the next collective
                except Exception as e:  # noqa: BLE001
                    ok_probe = 0
                    log(f"p2p probe: {e}")
                same = torch.tensor([ok_probe], device=ctrl)
                dist.all_reduce(same, op=dist.ReduceOp.MIN)

                if int(same.item()) == 1:
                    t_rccl, t_p2p = timed(step_rccl), timed(step_p2p)
                    t_best = min(t_best, t_rccl) if t_best is not None else t_rccl
                    # second comparison AFTER the buffers have been through several steps (a stale cache line in the
                    # peer-written result buffer would show here, not on first touch)
                    o_a, f_a = step_rccl(5)
                    o_b, f_b = step_p2p(5)
                    same = torch.tensor([int(torch.equal(o_a, o_b) and torch.equal(f_a, f_b))], device=ctrl)
                    dist.all_reduce(same, op=dist.ReduceOp.MIN)
                    log(f"transport probe: rccl {t_rccl * 1e3:.3f} ms/step, p2p {t_p2p * 1e3:.3f} ms/step, re-check {'ok' if int(same.item()) else 'MISMATCH'}")
                    p2p_verified = int(same.item()) == 1
                    if int(same.item()) == 1 and (args.transport == "p2p" or t_p2p < t_best):
                        step, transport, native, carrier, peer_ctxs = step_p2p, "peer-mapped stores (no all-to-all)", None, "p2p", [peer]
                        pd = max(1, args.p2p_depth)
                        if pd > 1 and not train:
                            # several lookups in flight: each owns a context (inboxes, result buffers, barrier flags) and a
                            # stream, so one lookup's partition / push / barrier run beside another's row traffic
                            try:
                                peers = [peer] + [PeerShardedFind(table, Router(world, batch, device=dev), max_batch=batch) for _ in range(pd - 1)]
                            except Exception as e:
                                peers = None
                                log(f"p2p depth {pd} unavailable: {e}")
                            if peers is not None:
                                pstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(pd - 1)]
                                for s_ in pstreams[1:]:
                                    s_.wait_stream(pstreams[0])

                                def step_p2p_pipe(i):
                                    with torch.cuda.stream(pstreams[i % pd]):
                                        return peers[i % pd].find(batches[i % n_batches], check_overflow=False, dedup=args.dedup)

                                o_c, f_c = step_p2p_pipe(1)      # runs on the second context
                                torch.cuda.synchronize(dev)
                                o_a, f_a = step_rccl(1)
                                same_p = torch.tensor([int(torch.equal(o_a, o_c) and torch.equal(f_a, f_c))], device=ctrl)
                                dist.all_reduce(same_p, op=dist.ReduceOp.MIN)
                                t_pipe = timed(step_p2p_pipe, k=8)
                                log(f"transport probe: p2p with {pd} lookups in flight {t_pipe * 1e3:.3f} ms/step, check {'ok' if int(same_p.item()) else 'MISMATCH'}")
                                if int(same_p.item()) == 1 and t_pipe < t_p2p:
                                    step, transport, peer_ctxs = step_p2p_pipe, f"peer-mapped stores (no all-to-all), {pd} lookups in flight", peers
                else:
                    log("p2p transport disagrees with the rccl path: not used")
        if train:   # the data-parallel training step: lookup, then every rank's gradients go to the owners, which apply
            grads = [torch.randn(batch, dim, device=dev) * 0.01 for _ in range(4)]
            find_step, via_peer = step, carrier == "p2p"

            def step(i):
                r_ = find_step(i)
                # --dedup: the backward sends ONE summed gradient row per distinct key of the rank's batch (mee_dedup_sum in front of the exchange; the native
                # context does it by its MEE_SHARDED_DEDUP flag)
                if via_peer:
                    peer.apply_adagrad(batches[i % n_batches], grads[i % 4], lr=0.01, eps=1e-10, check_overflow=False, dedup=args.dedup)
                elif native is not None:
                    native.apply_adagrad(batches[i % n_batches], grads[i % 4], lr=0.01, eps=1e-10)
                else:
                    shs[0].apply_adagrad(batches[i % n_batches], grads[i % 4], lr=0.01, eps=1e-10, dedup=args.dedup)
                return r_
        # the torch.distributed path end to end: what the timed region is re-run with if the chosen transport fails its check there
        def step_fallback(i):
            r_ = step_rccl(i)
            if train:
                shs[0].apply_adagrad(batches[i % n_batches], grads[i % 4], lr=0.01, eps=1e-10, dedup=args.dedup)
            return r_
    elif train:
        grads = [torch.randn(batch, dim, device=dev) * 0.01 for _ in range(4)]   # N(0, 1e-2), SURVEY §8d config 3

        slots = torch.empty(batch, dtype=torch.int64, device=dev)

        def step(i):   # forward gather (its launch also partitions the batch for the backward), then the backward scatter-update on the located slots
            o_, f_, _ = table.find_located(batches[i % n_batches], out=out, found=found, slots=slots, prepare_apply=True)
            table.apply_adagrad(batches[i % n_batches], grads[i % 4], lr=0.01, eps=1e-10, slots=slots)
            return o_, f_
    else:
        if n_out > 1:
            # a caller whose result buffers rotate knows that nothing re-reads them from cache: MEE_FIND_STREAM_STORES with every call is its hint
            # — the library's own rule only sees one call's size (64 MB: cached stores) — and one whose keys are uniform over a table far larger
            # than the caches adds MEE_FIND_STREAM_ROWS (skewed streams re-read their hot rows and lose with it: 27 -> 30 us on Zipf(1.05)).
            # The hints travel with the CALL (round 3 set them on the table, which other callers of the table share): keep whichever is fastest.
            from meepoembedding_amd import _lib as _ml
            choices = {"auto": None, "streaming_stores": _ml.FIND_STREAM_STORES, "streaming_stores_and_row_loads": _ml.FIND_STREAM_STORES | _ml.FIND_STREAM_ROWS}
            probe = {}
            for _round in range(2):   # every candidate twice, interleaved (the first one measured pays for cold caches and clocks otherwise): the better of its two
                for label, fl in choices.items():
                    us_ = kernel_window(table, batches, outs, founds, dev, launches=100, regions=3, warm=10, flags=fl)[0]
                    probe[label] = min(probe.get(label, us_), us_)
            best = min(probe, key=probe.get)
            find_flags = choices[best]
            store_hint = {"auto_us": probe["auto"], "streaming_stores_us": probe["streaming_stores"],
                          "streaming_stores_and_row_loads_us": probe["streaming_stores_and_row_loads"], "mee_find_ex_flags": find_flags or 0,
                          "used": "library default (mee_find)" if best == "auto" else f"{best.replace('_', ' ')} (mee_find_ex flags = {find_flags}, per call)"}
            log("rotating output buffers: " + ", ".join(f"{k} {v:.2f} us" for k, v in probe.items()) + " per launch")

        def step(i):
            return table.find(batches[i % n_batches], out=outs[i % n_out], found=founds[i % n_out], flags=find_flags)

    for i in range(args.warmup):
        r = step(i)
    torch.cuda.synchronize(dev)
    # correctness guard on the last warm-up batch: every looked-up key was inserted, rows are key-derived
    chk_keys = batches[(args.warmup - 1) % n_batches][:4096] if args.warmup else None
    if chk_keys is not None:
        o_rows, o_found = r
        assert bool(o_found.all()), "bench lookup missed an inserted key"
        if not train:   # in train mode the rows have been updated by earlier warm-up steps
            assert torch.equal(o_rows[:4096], synth.rows_t(chk_keys, dim, 2)), "bench lookup returned wrong rows"

    # ---- the timed region: EXACTLY --steps steps between barrier + synchronize pairs --------------------------------------
    # N=1 find mode issues the K steps as ONE hipGraph replay (K kernel nodes chained on one stream: the same launches, the
    # same in-order semantics, without K host-side launch calls in the region); --launch eager keeps one host call per step.
    graph = None
    launch_mode = "eager (one host launch call per step)"
    if args.launch == "graph" and not sharded and not train and args.steps > 0:
        try:
            gs_ = torch.cuda.Stream(dev)
            gs_.wait_stream(torch.cuda.current_stream(dev))
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=gs_):
                for i in range(args.steps):
                    step(i)
            g_.replay()                       # one untimed replay: upload + first-run costs stay out of the region
            torch.cuda.synchronize(dev)
            graph, launch_mode = g_, f"hipGraph: the {args.steps} steps captured as {args.steps} chained kernel nodes, one replay"
        except Exception as e:  # noqa: BLE001
            log(f"hipGraph capture unavailable ({e!r}): eager launches")
            graph = None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_region():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0_ = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for i_ in range(args.steps):
                step(i_)
        ev1.record()
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
        el_ = time.perf_counter() - t0_
        if sharded:
            tt_ = torch.tensor([el_], dtype=torch.float64, device=ctrl)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            el_ = float(tt_.item())
        return el_, ev0.elapsed_time(ev1)

    def transport_verdict():
        """collective: 1 only if the transport that carried the timed steps reports no overflow / time-out on ANY rank (the peer-mapped
        steps run with check_overflow=False, the padded native layout drops keys beyond a segment's capacity and raises a status bit)"""
        ok_ = 1
        if carrier == "native" and native.status() != 0:
            ok_ = 0
            print(f"[bench] rank {rank}: a padded segment of the native RCCL exchange overflowed during the timed steps", file=sys.stderr, flush=True)
        if carrier == "p2p":
            try:
                for pc in peer_ctxs:
                    pc.check()
            except Exception as e:  # noqa: BLE001
                ok_ = 0
                print(f"[bench] rank {rank}: peer-mapped transport reported {e} after the timed steps", file=sys.stderr, flush=True)
        if os.environ.get("MEE_BENCH_FAIL_TRANSPORT_CHECK") == "1":   # test hook (tests/test_bench_contract.py): rehearse the fallback below
            ok_ = 0
        okt = torch.tensor([ok_], dtype=torch.int32, device=ctrl)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        return int(okt.item())

    elapsed, ev_ms = timed_region()
    if sharded and carrier != "rccl" and transport_verdict() == 0:
        # a measured number must not come from steps that dropped keys or timed out: every rank switches to the torch.distributed
        # path and the region is run again (an automated N = 2 / 4 / 8 sweep still gets its line, labelled with what happened)
        log(f"transport '{transport}' failed its check during the timed steps: re-running them over torch.distributed")
        failed = transport
        step, native, carrier = step_fallback, None, "rccl"
        transport = f"rccl all-to-all (torch.distributed) [fallback: '{failed}' overflowed or timed out during the timed steps]"
        for i in range(max(args.warmup, 2)):
            step(i)
        torch.cuda.synchronize(dev)
        elapsed, ev_ms = timed_region()

    # ---- N > 1: SURVEY 8d config 4 (iii) in the SAME line: the uniform stream and Zipf(1.05) without and with pre-exchange dedup, each over every
    # carrier that can run it (verified against the torch.distributed path in the probe), best one reported with its bytes on the busiest link.
    # Every rank takes part (collectives inside); short windows (3 + 6 steps per candidate): well under a second per row.
    dist_streams = None
    if sharded and world > 1 and not train and not args.no_streams:
        dist_streams = {}
        if verified_native and not args.dedup:
            # the exchange behind the C-ABI de-duplicates per CONTEXT (MEE_SHARDED_DEDUP): one more context for the dedup row, checked against the
            # torch.distributed path's dedup lookup on a skewed batch before it may carry a number
            try:
                from meepoembedding_amd.sharded import RcclShardedTable
                ntd = RcclShardedTable(table, batch, pad_slack=0.0, dedup=True)
                native_tables.append(ntd)
                zb = lookup_batches(synth, n_keys, batch, 1, "zipf", dev, seed=3 + rank)[0]
                o_a, f_a = shs[0].find(zb, dedup=True)
                o_b, f_b = ntd.find(zb)
                okd = torch.tensor([int(torch.equal(o_a, o_b) and torch.equal(f_a, f_b) and ntd.status() == 0)], dtype=torch.int32, device=ctrl)
                dist.all_reduce(okd, op=dist.ReduceOp.MIN)
                if int(okd.item()) == 1:
                    verified_native.append(("exact segments, pre-exchange dedup", ntd))
            except Exception as e:  # noqa: BLE001
                log(f"native rccl with dedup unavailable: {e}")
        for sname, dname, dd in (("uniform", "uniform", False), ("zipf_1.05", "zipf", False), ("zipf_1.05_dedup", "zipf", True)):
            bs_ = batches if dname == args.dist else lookup_batches(synth, n_keys, batch, 8, dname, dev, seed=3 + rank)
            cands = [("rccl all-to-all (torch.distributed)", "rccl", 0, lambda i, bs_=bs_, dd=dd: shs[0].find(bs_[i % len(bs_)], dedup=dd), lambda: True)]
            for label_, nt_ in verified_native:
                if bool(getattr(nt_, "dedup", False)) == dd:
                    cands.append((f"RCCL grouped send/recv behind the C-ABI ({label_})", "native", nt_.segment_capacity,
                                  lambda i, bs_=bs_, nt_=nt_: nt_.find(bs_[i % len(bs_)]), lambda nt_=nt_: nt_.status() == 0))
            if p2p_verified and peer is not None:
                def p2p_ok_(peer=peer):
                    try:
                        peer.check()
                        return True
                    except Exception:  # noqa: BLE001  (an inbox overflowed: a skewed stream without dedup)
                        return False
                cands.append(("peer-mapped stores (no all-to-all)", "p2p", 0,
                              lambda i, bs_=bs_, dd=dd: peer.find(bs_[i % len(bs_)], check_overflow=False, dedup=dd), p2p_ok_))
            best = None
            for label_, carrier_, cap_, fn_, ok_fn in cands:
                t_c = timed(fn_)
                okc = torch.tensor([int(bool(ok_fn()))], dtype=torch.int32, device=ctrl)   # no overflow / time-out on ANY rank, or the number does not count
                dist.all_reduce(okc, op=dist.ReduceOp.MIN)
                if carrier_ == "native" and int(okc.item()) == 0:
                    for _, nt_ in verified_native:
                        nt_.clear_status() if hasattr(nt_, "clear_status") else None
                log(f"streams[{sname}]: {label_}: {t_c * 1e3:.3f} ms/step{'' if int(okc.item()) else ' (overflowed: not counted)'}")
                if int(okc.item()) == 1 and (best is None or t_c < best[0]):
                    best = (t_c, label_, carrier_, cap_)
            t_c, label_, carrier_, cap_ = best
            # bytes on the links: with dedup only a batch's DISTINCT keys (and their rows) travel
            traffic_b = [torch.unique(b_) for b_ in bs_[:1]] if dd else bs_
            busiest_, total_ = link_traffic(shs[0].router, traffic_b, carrier_, cap_, dim, world, ctrl)
            uq = torch.tensor([float(torch.unique(bs_[0]).numel()) / bs_[0].numel()], dtype=torch.float64, device=ctrl)
            dist.all_reduce(uq, op=dist.ReduceOp.SUM)
            dist_streams[sname] = {"value": world * batch / t_c, "unit": "key-lookups/s", "ms_per_step": t_c * 1e3, "transport": label_, "pre_exchange_dedup": dd,
                                   "unique_fraction_per_rank_batch": float(uq.item()) / world,
                                   "xgmi": {"bytes_on_busiest_link": busiest_, "frac": busiest_ / t_c / 1e9 / XGMI_LINK_GBS_PER_DIR, "total_bytes_per_step": total_,
                                            "aggregate_GBps": total_ / t_c / 1e9}}
        dist_streams["note"] = ("each row: 3 warm-up + 6 timed steps (barrier + synchronize on both sides, max over ranks) of the sharded find on that key stream over every "
                                "carrier verified against the torch.distributed path; the fastest one without a segment / inbox overflow is reported")

    launch_cmp = None
    if not sharded and not train and args.steps > 0:
        # the timed steps really ran: the LAST step's result buffer holds the rows of the last step's batch (a graph replay that skipped
        # nodes, or wrote elsewhere, fails here)
        last = args.steps - 1
        lk = batches[last % n_batches]
        assert bool(founds[last % n_out].all()) and torch.equal(outs[last % n_out][-4096:], synth.rows_t(lk[-4096:], dim, 2)), \
            "the timed steps did not leave the last batch's rows in its output buffer"
        if graph is not None:   # the same K steps issued eagerly (one host launch call per step), HIP events: reported beside the graph number
            torch.cuda.synchronize(dev)
            ev0.record()
            for i_ in range(args.steps):
                step(i_)
            ev1.record()
            torch.cuda.synchronize(dev)
            launch_cmp = {"graph_us_per_step": ev_ms * 1e3 / args.steps, "eager_us_per_step": ev0.elapsed_time(ev1) * 1e3 / args.steps,
                          "graph_wall_us_per_step": elapsed / args.steps * 1e6}

    # dominant kernel (find_kernel) alone in its own fixed window (5 x 200 launches, HIP events on the launch stream): the
    # roofline object does not depend on --steps
    roof_n = batch          # lookups per launch of the kernel the roofline object prices
    xgmi = None
    if train and not sharded:
        kern_s = ev_ms / 1e3 / args.steps
        kern_min_s = kern_s
    elif sharded and world > 1:
        # the local find_kernel as the sharded step runs it: over keys THIS shard owns (what arrives from the G sources), about `batch` of them
        mine_ = torch.cat([b_[hash_batch(b_, 1, world)[2] == rank] for b_ in batches])
        kb = [mine_[s_:s_ + batch] for s_ in range(0, mine_.numel() - batch + 1, batch)] or [mine_]
        roof_n = kb[0].numel()
        med_us, min_us = kernel_window(table, kb, out[:roof_n], found[:roof_n], dev, launches=50, regions=3, warm=5)
        kern_s, kern_min_s = med_us / 1e6, min_us / 1e6
        busiest, total_b = link_traffic(shs[0].router, batches, carrier, native.segment_capacity if (carrier == "native" and native is not None) else 0,
                                        dim, world, ctrl, payload_dim=dim if train else 0)
        t_step = elapsed / args.steps
        xgmi = {"bytes_on_busiest_link": busiest, "link_peak_GBps_per_direction": XGMI_LINK_GBS_PER_DIR,
                "frac": busiest / t_step / 1e9 / XGMI_LINK_GBS_PER_DIR, "total_bytes_per_step": total_b, "aggregate_GBps": total_b / t_step / 1e9,
                "links_are_xgmi": args.backend == "nccl",
                "note": "algorithmic bytes per step on the busiest ordered GPU pair (keys out + rows and found bytes back; padded segments at their "
                        "constant size), from the owner counts of a representative batch; fraction of the nominal 76.8 GB/s per link direction (SURVEY 8e)"}
    else:
        med_us, min_us = kernel_window(table, batches, outs, founds, dev, flags=find_flags)   # as the timed steps run: rotating output buffers, the caller's per-call hints
        kern_s, kern_min_s = med_us / 1e6, min_us / 1e6
    bpl = algorithmic_bytes_per_lookup(dim)
    reused_us = None
    if not sharded and not train and n_out > 1:   # the same launches into ONE reused 64 MB buffer under the library's default policy (what round 2 reported)
        reused_us = kernel_window(table, batches, out, found, dev)
    whole = train and not sharded   # sharded runs always price the local find_kernel alone
    if whole:
        # fwd 528 B/lookup + bwd 264 B/lookup + 1032 B per unique key (Adagrad); here the whole step is priced
        uniq = sum(int(torch.unique(b_).numel()) for b_ in batches[:8]) / 8
        step_bytes = (bpl + 8 + 4 * dim) * batch + (8 + 16 * dim) * uniq
        achieved = step_bytes / kern_s / 1e9
    else:
        achieved = roof_n * bpl / kern_s / 1e9

    if rank == 0:
        value = world * batch * args.steps / elapsed
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "find_traffic.json")  # written by tools/pmc_traffic.py from rocprofv3 --pmc passes
        if os.path.exists(tpath) and not sharded:
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == batch and tj.get("dim") == dim and tj.get("keys") == keys_per_gpu:
                    traffic = tj["bytes_per_launch"]   # measured in separate rocprofv3 --pmc passes (tools/pmc_traffic.py), not in this run
            except Exception:
                traffic = None
        step_traffic = step_traffic_src = None   # configs[2]: the step's kernels summed, from the PMC passes of the latest evidence run (tools/final_r5.sh)
        apath = os.path.join(ROOT, "profiles", "step_traffic.json")
        if whole and not sharded and os.path.exists(apath) and args.dist == "uniform" and dim == 64 and keys_per_gpu == 100_000_000:
            try:
                aj = json.load(open(apath))
                if aj.get("batch") == batch:
                    step_traffic = sum(k_["bytes_per_launch"] for k_ in aj["kernels"].values())
                    step_traffic_src = ("profiles/step_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/final_r5.sh; "
                                        "the step's kernels summed: " + ", ".join(aj["kernels"]) + ")")
            except Exception:
                step_traffic = None
        res = {
            "metric": "key-lookups/sec" if not train else "train-step keys/sec (find + sparse Adagrad apply)", "value": value, "unit": "key-lookups/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64 keys / fp32 rows (copy)", "data": "synthetic",
            "config": {"workload": (f"row-sharded {'train step (find + gradient exchange + sparse Adagrad)' if train else 'find'}: {n_keys // 1_000_000}M keys over {world} GPUs ({keys_per_gpu // 1_000_000}M/GPU), dim {dim}, "
                                    f"{batch} lookups per rank per step, transport: {transport}" if sharded else
                                    f"configs[2]: 1xMI355X, {n_keys // 1_000_000}M keys, dim {dim}, forward find + sparse-Adagrad scatter-update, {batch}-key batches" if train else
                                    f"configs[1]: 1xMI355X, {n_keys // 1_000_000}M keys, dim {dim} fp32, forward find only, {batch}-key batches"
                                    + (f", results rotating over {n_out} output buffers ({n_out * batch * dim * 4 >> 20} MB > Infinity Cache)" if n_out > 1 else ", one reused output buffer")),
                       "keys_per_gpu": keys_per_gpu, "local_size": local_size, "dim": dim, "batch_per_rank": batch, "load_factor": args.load,
                       "key_distribution": args.dist, "table_gb": round(table.table_bytes / 1e9, 2), "launch": launch_mode,
                       "launch_comparison": launch_cmp, "output_buffers": (f"{n_out} x {batch * dim * 4 >> 20} MB dense result buffers in rotation" if n_out > 1 else "one reused result buffer"),
                       "parallelism": (f"row-shard x{world}" + (f", {args.pipeline} steps in flight" if args.pipeline > 1 else "") + (", pre-exchange dedup" + (" (lookup keys and summed gradient rows)" if train else "") if args.dedup else "")) if sharded else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic if not whole else step_traffic, "kernel": "find_kernel" if not whole else "whole step (find_prepare_kernel: the located find + the apply's partition; bkt_apply_kernel: dedup + update)",
                         "traffic_source": (("profiles/find_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE x2 per the gfx950 correction)"
                                             if traffic is not None else None) if not whole else step_traffic_src),
                         "avg_launch_us": kern_s * 1e6, "min_launch_us": kern_min_s * 1e6,
                         # N=1 find: `frac` prices the launches as the timed steps run them — rotating over n_out result buffers (more than the
                         # Infinity Cache holds); the same launches into ONE reused buffer are reported beside it
                         "frac_out_rotating": (achieved / HBM_PEAK_GBS) if reused_us is not None else None,
                         # ... and what a caller that passes NO hint gets into the same rotating buffers (the library's per-call rule: mee_find)
                         "frac_library_default": (batch * bpl / store_hint["auto_us"] / 1e3 / HBM_PEAK_GBS) if store_hint is not None else None,
                         "frac_out_reused": (batch * bpl / reused_us[0] / 1e3 / HBM_PEAK_GBS) if reused_us is not None else None,
                         "out_reused_avg_launch_us": reused_us[0] if reused_us is not None else None,
                         "out_buffers": n_out, "out_store_policy": store_hint,
                         "window": "median of 5 HIP-event windows of 200 back-to-back launches on the launch stream (independent of --steps)" if not whole else "the timed steps",
                         "algorithmic_bytes_per_lookup": bpl if not whole else step_bytes / batch,
                         "read_only_GBps": roof_n * (16 + 4 * dim) / kern_s / 1e9 if not whole else None,
                         "lookups_per_launch": roof_n},
        }
        if dist_streams is not None:
            res["streams"] = dist_streams
        if xgmi is not None:
            res["xgmi"] = xgmi
            res["roofline"]["window"] = ("median of 3 HIP-event windows of 50 back-to-back launches of the LOCAL find_kernel over keys this shard owns "
                                         "(what arrives from the G sources per step), on rank 0")
        if not sharded and not train and not args.no_streams:
            try:
                res["streams"] = stream_table(table, synth, n_keys, batch, dim, dev, out, found, bpl, batches, reused_us if reused_us is not None else (kern_s * 1e6, kern_min_s * 1e6))
                res["streams"]["note"] = "every row but north_star_batch_1M and the two_output_buffers rows: launches into ONE reused result buffer, default cache policy"
            except Exception as e:  # noqa: BLE001
                res["streams"] = {"error": repr(e)}
        if whole and not args.no_streams:
            try:
                res["streams"] = train_streams(table, synth, n_keys, batch, dim, dev, out, found, bpl)
            except Exception as e:  # noqa: BLE001
                res["streams"] = {"error": repr(e)}
        if not sharded and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(synth, dim, batch, keys_per_gpu, args.load)
        if not sharded and not train and args.extras:
            try:  # never allowed to break the headline line
                res["also"] = {"two_caller_streams": two_stream_extra(table, batches, dim, dev, bpl)}
            except Exception as e:  # noqa: BLE001
                res.setdefault("also", {})["error"] = repr(e)
        if not sharded and not train and not args.no_streams and not args.no_configs2:
            try:  # configs[2] beside the headline, in the line the driver records (never allowed to break it)
                del outs, founds
                res["configs2"] = configs2_rows(table, synth, n_keys, dim, dev, chunk, batch, bpl)
            except Exception as e:  # noqa: BLE001
                res["configs2"] = {"error": repr(e)}
        if not sharded and args.tier_1b:
            try:
                table.close()
                del batches
                torch.cuda.empty_cache()
                res.setdefault("also", {})["tier_1b_keys_one_gpu"] = tier_1b_point(synth, dim, dev, batch, log)
            except Exception as e:  # noqa: BLE001
                res.setdefault("also", {})["tier_1b_keys_one_gpu"] = {"error": repr(e)}
        if not sharded and args.tier_share:
            try:
                table.close()
                batches = None
                torch.cuda.empty_cache()
                res.setdefault("also", {})["tier_share_one_gpu"] = tier_share_point(synth, dim, dev, 1 << 20, log, *[int(x) for x in args.tier_share_keys.split(",")])
            except Exception as e:  # noqa: BLE001
                res.setdefault("also", {})["tier_share_one_gpu"] = {"error": repr(e)}
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if sharded:
        torch.cuda.synchronize(dev)
        dist.barrier()
        for nt_ in native_tables:   # every rank destroys its communicators while all peers are still alive
            nt_.close()
        if peer is not None:
            try:
                peer.close()
            except Exception:  # noqa: BLE001
                pass
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""rocprofv3 --pmc CSVs of tools/pmc_sq.sh -> a markdown table per key stream: per kernel the median of every counter over its dispatches, and
the ratios that say where the wave-time goes (SQ counters are summed over all waves of a dispatch; *_CYCLES in quad-cycles: MI355X_MICROARCH.md).
usage: python tools/pmc_sq_summary.py gpurun_out/pmc_sq > profiles/r04_apply_sq.md"""
import csv, glob, os, statistics, sys
root = sys.argv[1]
KERNELS = ["bkt_sort_kernel", "bkt_apply_kernel<1, 16, false, false, false>", "bkt_apply_kernel<1, 16, true, false, false>", "bkt_apply_kernel<1, 16, false, false, true>",
           "bkt_apply_kernel<1, 16, true, false, true>", "find_kernel<16, 2, 64>", "find_prepare_kernel<16, 2, 64>"]
print("# SQ counters of the apply path's kernels (round 4; `…, false>` = the LEAN kernel of uniform streams, `…, true>` = the FULL kernel of skewed streams)\n")
print("`tools/pmc_sq.sh`: one `rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE "
      "--kernel-trace` pass per key stream over `tools/apply_trace.py 100000000 <stream>` (100M keys, dim 64, 256K-key batches). "
      "Medians over a kernel's dispatches.  SQ counters are sums over all waves; `wait` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave-time parked on s_waitcnt / barriers), "
      "`issue` = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, `stall` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, `waves in flight` = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES per SE-summed busy) "
      "is not derivable without per-SE data and is left out; `occupancy` below = mean resident waves per CU = 4 x SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE / 256 CUs.\n")
for stream in ("uniform", "zipf"):
    rows = {}
    for f in glob.glob(os.path.join(root, stream, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    rows.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(f"## {stream} keys\n")
    print("| kernel | dispatches | waves | VMEM rd / wave | VMEM wr / wave | wait | issue | stall | mean resident waves / CU |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k in KERNELS:
        if k not in rows:
            continue
        m = {c: statistics.median(v) for c, v in rows[k].items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        waves = m.get("SQ_WAVES", 0) or 1
        occ = 4 * wc / (m.get("GRBM_GUI_ACTIVE", 0) or 1) / 256
        print(f"| `{k}` | {len(rows[k].get('SQ_WAVES', []))} | {waves:.0f} | {m.get('SQ_INSTS_VMEM_RD', 0) / waves:.1f} | {m.get('SQ_INSTS_VMEM_WR', 0) / waves:.1f} | "
              f"{m.get('SQ_WAIT_ANY', 0) / wc:.2f} | {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {occ:.1f} |")
    print()

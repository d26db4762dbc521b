"""Parse rocprofv3 --pmc CSV output (separate FETCH_SIZE / WRITE_SIZE passes) into per-launch HBM traffic.

usage: python tools/pmc_traffic.py <kernel-substring> <dir-with-FETCH_SIZE-pass> <dir-with-WRITE_SIZE-pass> [--calib-read X] [--out file.json] [--meta k=v ...]
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md §HBM).  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so the
guide prescribes doubling it for wide coalesced reads; --calib-read overrides that factor with one measured on a kernel of
known byte count in the same access pattern (tools/gather_ubench2: 264 B read + 256 B written per key).
"""
import argparse, csv, glob, json, os, statistics, sys


def counter_per_dispatch(d, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel"); ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--calib-read", type=float, default=2.0)
    ap.add_argument("--out"); ap.add_argument("--meta", nargs="*", default=[])
    a = ap.parse_args()
    f = counter_per_dispatch(a.fetch_dir, a.kernel, "FETCH_SIZE"); w = counter_per_dispatch(a.write_dir, a.kernel, "WRITE_SIZE")
    if not f or not w:
        sys.exit(f"no rows for kernel '{a.kernel}' (fetch {len(f)}, write {len(w)})")
    fk, wk = statistics.median(f), statistics.median(w)
    res = {"kernel": a.kernel, "dispatches": [len(f), len(w)], "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
           "read_correction": a.calib_read, "read_bytes_per_launch": fk * 1024 * a.calib_read, "write_bytes_per_launch": wk * 1024,
           "bytes_per_launch": fk * 1024 * a.calib_read + wk * 1024}
    for kv in a.meta:
        k, v = kv.split("="); res[k] = int(v) if v.isdigit() else v
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)

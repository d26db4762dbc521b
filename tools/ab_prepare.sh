#!/bin/bash
# usage (on the GPU box): tools/ab_prepare.sh OUTDIR DEBUGVALUE… — the training forward under "prepare_debug" values: step times + per-kernel averages
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== prepare_debug $v" >> $out/prep.txt
  rm -rf /tmp/abp && MEE_PREPARE_DEBUG=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o x -- python3 tools/apply_trace.py 100000000 uniform 1 2>&1 | grep apply_path >> $out/prep.txt || exit 1
  python3 - >> $out/prep.txt <<'PY'
import csv, glob
f = glob.glob("/tmp/abp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "mee::" in r["Name"] and any(k in r["Name"] for k in ("bkt_", "find_prepare", "find_kernel")):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
done
cat $out/prep.txt

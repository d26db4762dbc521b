#!/bin/bash
# usage: ab_uniform.sh OUT LIB... : located/probing apply kernel time on uniform keys per library
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export MEE_LIB_PATH=$GRAFT_REPO_ROOT/$lib; else unset MEE_LIB_PATH; fi
  tag=$(basename "$lib" .so)
  rm -rf /tmp/abp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o x -- python3 tools/apply_trace.py 100000000 uniform 1 > /dev/null 2>&1 || exit 1
  echo "== $tag uniform" >> $out/kernels.txt
  python3 - >> $out/kernels.txt <<'PY'
import csv, glob
f = glob.glob("/tmp/abp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "mee::" in r["Name"] and any(k in r["Name"] for k in ("bkt_apply",)):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
done
cat $out/kernels.txt

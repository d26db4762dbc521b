#!/bin/bash
# usage (on the GPU box): tools/ab_apply.sh OUTDIR LIB… — for each library ("" = the in-tree one): apply_trace uniform + zipf, then per-kernel
# averages of the zipf run from rocprofv3 --kernel-trace --stats
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export MEE_LIB_PATH=$GRAFT_REPO_ROOT/$lib MEE_LIB_OLDER_BUILD=1; else unset MEE_LIB_PATH MEE_LIB_OLDER_BUILD; fi
  tag=$(basename "$lib" .so)
  echo "== $tag" >> $out/times.txt
  for d in uniform zipf; do timeout -k 10 120 python3 tools/apply_trace.py 100000000 $d 1 2>&1 | grep apply_path >> $out/times.txt || exit 1; done
  for d in uniform zipf; do
    rm -rf /tmp/abp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o x -- python3 tools/apply_trace.py 100000000 $d 1 > /dev/null 2>&1 || exit 1
    echo "== $tag $d" >> $out/kernels.txt
    python3 - >> $out/kernels.txt <<'PY'
import csv, glob
f = glob.glob("/tmp/abp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "mee::" in r["Name"] and any(k in r["Name"] for k in ("bkt_", "find_prepare", "find_kernel")):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
  done
done
cat $out/times.txt $out/kernels.txt

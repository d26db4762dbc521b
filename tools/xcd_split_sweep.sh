#!/bin/bash
# usage (GPU box): tools/xcd_split_sweep.sh OUT SPLIT… — apply_trace uniform (+ zipf) per value of the "apply_xcd_split" knob (0 = even halves), un-profiled times and rocprofv3 kernel averages
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in "$@"; do
  export MEE_XCD_SPLIT=$sp
  echo "== split $sp" >> $out/sweep.txt
  for d in uniform zipf; do
    rm -rf /tmp/xs; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xs -o x -- python3 tools/apply_trace.py 100000000 $d > /tmp/xs.log 2>&1
    grep "per step" /tmp/xs.log | sed "s/apply_path 1 //" >> $out/sweep.txt
    python3 tools/kernel_stats.py /tmp/xs bkt_apply >> $out/sweep.txt
  done
done
cat $out/sweep.txt

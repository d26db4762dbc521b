#!/bin/bash
# usage (GPU box): bash tools/dedup_kernels.sh OUT.txt [LIB] — dedup_keys / assign per 1M keys and their kernels, per key stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$2" ] && export MEE_LIB_PATH=$GRAFT_REPO_ROOT/$2
for d in uniform zipf; do
  echo "== $d" >> $1
  MEE_DEDUP_DIST=$d timeout -k 10 200 python3 tools/dedup_bench.py 2>&1 | grep "us per" >> $1 || exit 1
  rm -rf /tmp/ddp && MEE_DEDUP_DIST=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ddp -o x -- python3 tools/dedup_bench.py > /dev/null 2>&1 || exit 1
  python3 tools/kernel_stats.py /tmp/ddp bkt_ assign_reserved >> $1
done
cat $1

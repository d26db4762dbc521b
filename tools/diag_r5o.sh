#!/bin/bash
# usage (GPU box): bash tools/diag_r5o.sh OUTDIR — the first skewed steps' dispatches in order; the sum kernel's phase timeline with sub-stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=$1; mkdir -p $o
rm -rf /tmp/fsb && MEE_FSB_SHORT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fsb -o x -- python3 tools/first_skewed_batch.py > $o/fsb.txt 2>&1 || exit 1
python3 tools/kernel_sequence.py /tmp/fsb -40 > $o/fsb_sequence.txt || exit 1
for d in uniform zipf; do
  MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d > $o/stl_$d.txt 2>&1 || exit 1
  MEE_STL_NULL=inverse MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d > $o/stl_${d}_noinverse.txt 2>&1 || exit 1
done
echo done

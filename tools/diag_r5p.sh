#!/bin/bash
# usage (GPU box): bash tools/diag_r5p.sh OUTDIR — after a kernel change: the suite, then the first skewed steps, dedup per 1M keys, the apply paths per stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=$1; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $o/tests.txt 2>&1; rc=$?; tail -3 $o/tests.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/first_skewed_batch.py 2>&1 | grep -v amdgpu > $o/first_skewed.txt || exit 1
cat $o/first_skewed.txt
rm -rf /tmp/fsb && MEE_FSB_SHORT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fsb -o x -- python3 tools/first_skewed_batch.py > /dev/null 2>&1 || exit 1
python3 tools/kernel_sequence.py /tmp/fsb -26 > $o/fsb_sequence.txt || exit 1
timeout -k 10 300 python3 tools/dedup_bench.py 2>&1 | grep "us" > $o/dedup.txt || exit 1
cat $o/dedup.txt
for d in uniform zipf; do timeout -k 10 300 python3 tools/apply_trace.py 100000000 $d 2>&1 | grep apply_path >> $o/apply.txt || exit 1; done
cat $o/apply.txt
for d in uniform zipf; do MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d > $o/stl_$d.txt 2>&1 || exit 1; done
echo done

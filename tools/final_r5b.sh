#!/bin/bash
# usage (GPU box): bash tools/final_r5b.sh OUTDIR — second evidence pass of round 5 (after the dedup_sum / first-skewed-batch work): suite, dedup per 1M keys with kernel rows,
# sum timeline, first skewed steps with their dispatches, apply paths, world-1 sharded steps with and without --dedup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=$1; mkdir -p $o
bash tools/gpu_suite.sh $o || exit 1
grep -q "pytest rc=0" $o/tests.txt || exit 1
bash tools/dedup_kernels.sh $o/dedup.txt > /dev/null || exit 1
for d in uniform zipf; do MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d 2>&1 | grep -v amdgpu > $o/stl_$d.txt || exit 1; done
timeout -k 10 300 python3 tools/first_skewed_batch.py 2>&1 | grep -v amdgpu > $o/first_skewed.txt || exit 1
rm -rf /tmp/fsb && MEE_FSB_SHORT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fsb -o x -- python3 tools/first_skewed_batch.py > /dev/null 2>&1 || exit 1
python3 tools/kernel_sequence.py /tmp/fsb -26 > $o/fsb_sequence.txt || exit 1
for d in uniform zipf; do timeout -k 10 300 python3 tools/apply_trace.py 100000000 $d 2>&1 | grep apply_path >> $o/apply.txt || exit 1; done
bash tools/world1_zipf.sh $o/world1 > $o/world1.txt 2>&1 || exit 1
echo done

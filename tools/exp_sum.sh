#!/bin/bash
# usage (GPU box): bash tools/exp_sum.sh NAME — after a change to the dedup kernels: their tests, times per 1M keys, the sum kernel's phase timeline (diagnostic build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o
timeout -k 10 240 python3 -m pytest tests/test_dedup_sum.py tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "dedup or assign or aggregat" > $o/tests.txt 2>&1; rc=$?; tail -2 $o/tests.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 150 python3 tools/dedup_bench.py 2>&1 | grep "us" >> $o/sum.txt || exit 1
cat $o/sum.txt
for d in uniform zipf; do MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d 2>&1 | grep -v amdgpu > $o/stl_$d.txt || exit 1; done
echo done

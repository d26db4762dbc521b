#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r5af; mkdir -p $o
bash tools/gpu_suite.sh $o || exit 1
grep -q "pytest rc=0" $o/tests.txt || exit 1
bash tools/world1_dedup_trace.sh $o/w1d.txt > /dev/null || exit 1
head -8 $o/w1d.txt
bash tools/world1_zipf.sh $o/world1 > $o/world1.txt 2>&1 || exit 1
cat $o/world1.txt
timeout -k 10 150 python3 tools/dedup_bench.py 2>&1 | grep "us" > $o/dedup.txt || exit 1
cat $o/dedup.txt

#!/bin/bash
# usage (GPU box): bash tools/stl_only.sh NAME [uniform|zipf ...] — the sum kernel's phase timeline (diagnostic build: build/libmeepo_hip_stl.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$1; mkdir -p $o; shift
for d in "${@:-zipf}"; do MEE_LIB_PATH=$GRAFT_REPO_ROOT/build/libmeepo_hip_stl.so timeout -k 10 200 python3 tools/sum_timeline.py $d 2>&1 | grep -v amdgpu > $o/stl_$d.txt || exit 1; cat $o/stl_$d.txt; done

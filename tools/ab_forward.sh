#!/bin/bash
# usage (GPU box): tools/ab_forward.sh OUT LIB… ("-" = the in-tree library): the training forward's kernels (find_located, find_located_prepare) per library, uniform keys
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export MEE_LIB_PATH=$GRAFT_REPO_ROOT/$lib MEE_LIB_OLDER_BUILD=1; else unset MEE_LIB_PATH MEE_LIB_OLDER_BUILD; fi
  tag=$(basename "$lib" .so)
  rm -rf /tmp/abp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o x -- python3 tools/apply_trace.py 100000000 uniform > /tmp/abf.log 2>&1
  echo "== $tag" >> $out/forward.txt
  grep "per step" /tmp/abf.log >> $out/forward.txt
  python3 tools/kernel_stats.py /tmp/abp find_prepare "find_kernel<16, 2, 64>" >> $out/forward.txt
done
cat $out/forward.txt

#!/bin/bash
# usage (GPU box): bash tools/world1_dedup_trace.sh OUT.txt — per-kernel times of the sharded training step at world 1 on a Zipf(1.05) stream with --dedup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/w1d && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w1d -o x -- python3 bench.py --mode train --force-sharded --dist zipf --dedup --no-cpu-baseline --no-streams --steps 40 --warmup 10 > /tmp/w1d.json 2> /tmp/w1d.err || { tail -n 20 /tmp/w1d.err; exit 1; }
python3 tools/kernel_stats.py /tmp/w1d mee:: > $1
python3 tools/kernel_sequence.py /tmp/w1d -60 bkt_ find_ scatter gather copy >> $1
cat $1

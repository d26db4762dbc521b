#!/bin/bash
# usage (GPU box): bash tools/native_dedup_w1.sh OUT.txt — the NATIVE sharded context (mee_sharded_*, RCCL at world 1) on a Zipf(1.05) stream: lookup and training step without / with MEE_SHARDED_DEDUP,
# and the kernels of the --dedup lookup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in find train; do for d in "" "--dedup"; do
  timeout -k 10 300 python3 bench.py --mode $m --force-sharded --transport native --dist zipf $d --no-cpu-baseline --no-streams --steps 60 --warmup 10 > /tmp/nd.json 2> /tmp/nd.err || { tail -n 20 /tmp/nd.err; exit 1; }
  python3 -c "import json; d=json.loads(open('/tmp/nd.json').read().strip().splitlines()[-1]); print('native world1 zipf mode=$m $d:', round(d['ms_per_step'], 4), 'ms per step;', d['config']['workload'][-120:])" >> $1
done; done
rm -rf /tmp/ndp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ndp -o x -- python3 bench.py --mode find --force-sharded --transport native --dist zipf --dedup --no-cpu-baseline --no-streams --steps 60 --warmup 10 > /dev/null 2>&1 || exit 1
python3 tools/kernel_stats.py /tmp/ndp shard_ bkt_ find_kernel permute part_ >> $1
cat $1

#!/bin/bash
# usage (GPU box): bash tools/round4_multi.sh OUTDIR — (1) the N = 4 launch of the driver's SCALE run rehearsed on ONE GPU over gloo with reduced
# tables and --verbose (per-phase wall times; --no-selftest: four ranks and their four self-test children would be eight processes on one
# card), (2) the same at N = 2 with the self-tests, (3) the sharded lookup at world 1 on a Zipf stream with and without the pre-exchange dedup
O=$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s); timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --backend gloo --keys 20000000 --no-selftest --verbose > $O/world4_gloo.json 2> $O/world4_gloo.err || { tail -n 30 $O/world4_gloo.err; exit 1; }
grep "^\[bench" $O/world4_gloo.err | cut -c1-200 | tail -n 30; echo "world 4: wall $(( $(date +%s) - T0 )) s"
T0=$(date +%s); timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --backend gloo --keys 20000000 --verbose > $O/world2_gloo.json 2> $O/world2_gloo.err || { tail -n 30 $O/world2_gloo.err; exit 1; }
grep "^\[bench" $O/world2_gloo.err | cut -c1-200 | tail -n 30; echo "world 2: wall $(( $(date +%s) - T0 )) s"
for d in "" "--dedup"; do
  timeout -k 10 300 python3 bench.py --force-sharded --dist zipf $d --no-cpu-baseline --steps 100 --warmup 20 > $O/world1_zipf$d.json 2> $O/world1_zipf$d.err || { tail -n 20 $O/world1_zipf$d.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/world1_zipf$d.json').read().strip().splitlines()[-1]); print('world1 zipf $d', d['value'], d['ms_per_step'])"
done

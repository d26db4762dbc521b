#!/bin/bash
# usage (GPU box): bash tools/round5_multi.sh OUTDIR — the driver's N > 1 launch rehearsed on ONE GPU over gloo with reduced tables and --verbose:
# (1) N = 5 — five ranks are the most the box admits on its card beside nothing else (six processes per job; --no-selftest: the transports' self-test
# children would be processes of their own) —, (2) N = 2 training step (lookup + gradient exchange + sparse Adagrad) on a Zipf(1.05) stream without and with
# the pre-exchange reductions (--dedup: distinct keys out, one summed gradient row per distinct key back to the owners)
O=$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s); timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 5 --backend gloo --keys 16000000 --no-selftest --verbose > $O/world5_gloo.json 2> $O/world5_gloo.err || { tail -n 30 $O/world5_gloo.err; exit 1; }
grep "^\[bench" $O/world5_gloo.err | cut -c1-200 | tail -n 30; echo "world 5: wall $(( $(date +%s) - T0 )) s"
for d in "" "--dedup"; do
  T0=$(date +%s); timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29553 bench.py --gpus 2 --backend gloo --keys 16000000 --mode train --dist zipf $d --no-selftest --verbose --steps 50 --warmup 10 > $O/world2_train_zipf$d.json 2> $O/world2_train_zipf$d.err || { tail -n 30 $O/world2_train_zipf$d.err; exit 1; }
  python3 -c "import json; d=json.loads(open('$O/world2_train_zipf$d.json').read().strip().splitlines()[-1]); print('world 2 train zipf $d:', d['value'], d['unit'], d['ms_per_step'], 'ms per step;', d['config']['workload'][:200])"
  echo "  wall $(( $(date +%s) - T0 )) s"
done

#!/bin/bash
# usage (GPU box): bash tools/world1_zipf.sh OUTDIR — the sharded lookup and the sharded training step (lookup + Adagrad backward) at world 1 on a Zipf(1.05) stream,
# without and with the pre-exchange reductions (distinct keys for the lookup, one summed gradient row per distinct key for the backward)
O=$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in find train; do for d in "" "--dedup"; do
  timeout -k 10 300 python3 bench.py --mode $m --force-sharded --dist zipf $d --no-cpu-baseline --steps 100 --warmup 20 > $O/world1_zipf_$m$d.json 2> $O/world1_zipf_$m$d.err || { tail -n 20 $O/world1_zipf_$m$d.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/world1_zipf_$m$d.json').read().strip().splitlines()[-1]); print('world1 zipf mode=$m $d:', d['value'], d['unit'], d['ms_per_step'], 'ms per step;', d['config']['workload'][:160])"
done; done

#!/bin/bash
# usage (GPU box): bash tools/world1_zipf.sh OUTDIR — the sharded lookup at world 1 on a Zipf(1.05) stream, with and without the pre-exchange dedup
O=$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in "" "--dedup"; do
  timeout -k 10 300 python3 bench.py --force-sharded --dist zipf $d --no-cpu-baseline --steps 100 --warmup 20 > $O/world1_zipf$d.json 2> $O/world1_zipf$d.err || { tail -n 20 $O/world1_zipf$d.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/world1_zipf$d.json').read().strip().splitlines()[-1]); print('world1 zipf $d', d['value'], d['ms_per_step'])"
done

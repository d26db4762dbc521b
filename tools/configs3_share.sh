#!/bin/bash
# configs[3]'s per-GPU share (1B keys / 8 GPUs = 125M keys, dim 128) on ONE MI355X: find at 256K- and 1M-key batches, the find + Adagrad step.
# usage (GPU box): bash tools/configs3_share.sh OUT.json
set -e
cd $GRAFT_REPO_ROOT
T=/tmp/c3 && rm -rf $T && mkdir -p $T
timeout -k 10 300 python3 bench.py --keys 125000000 --dim 128 --no-cpu-baseline --no-configs2 > $T/find_256K.json 2> $T/find_256K.err
timeout -k 10 300 python3 bench.py --keys 125000000 --dim 128 --batch 1048576 --out-buffers 3 --no-cpu-baseline --no-streams --no-configs2 > $T/find_1M.json 2> $T/find_1M.err
timeout -k 10 300 python3 bench.py --keys 125000000 --dim 128 --mode train --no-cpu-baseline > $T/train.json 2> $T/train.err
python3 - "$1" <<'PY'
import json, sys
lines = {k: json.loads(open(f"/tmp/c3/{k}.json").read().strip().splitlines()[-1]) for k in ("find_256K", "find_1M", "train")}
json.dump({"what": "configs[3]'s per-GPU share on ONE MI355X: 125M keys, dim 128 (bench.py --keys 125000000 --dim 128 [--batch 1048576 | --mode train])",
           "lines": lines}, open(sys.argv[1], "w"), indent=1)
for k, v in lines.items():
    print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"])
PY

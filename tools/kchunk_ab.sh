# same-box A/B of library builds on the apply path: VARIANTS="a b" -> build/libmeepo_a.so ...; DISTS="zipf uniform"
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for v in ${VARIANTS:-prev new}; do
 for d in ${DISTS:-zipf}; do
  echo "== $v $d" >> gpurun_out/kchunk_ab.txt
  MEE_LIB_PATH=build/libmeepo_$v.so timeout -k 10 200 python3 tools/tune_apply.py --dist $d --rounds 3 2>&1 | grep -E "^adagrad|round 1" >> gpurun_out/kchunk_ab.txt
  MEE_LIB_PATH=build/libmeepo_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc_$v -o run -- python3 tools/apply_trace.py 100000000 $d > /dev/null 2>&1
  python3 - <<PY >> gpurun_out/kchunk_ab.txt
import csv,glob
f=glob.glob("gpurun_out/kc_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('group_kernel','apply_')): print(f"  {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
 done
done

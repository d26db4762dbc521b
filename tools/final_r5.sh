# end-of-round evidence run on the GPU box (round 5): bench lines + rocprofv3 kernel stats + PMC traffic + per-kernel tables -> gpurun_out/final5/
# usage: bash tools/final_r5.sh [part]   part 1 = bench lines, kernel stats, PMC traffic; part 2 = apply traces, SQ counters, dedup, first skewed batches, calibration;
#                                        part 3 = all-ops rooflines, world-1 sharded Zipf, timelines
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final5
O=gpurun_out/final5
R=/tmp/mee_final5_raw   # raw rocprofv3 output (kernel traces, counter CSVs: hundreds of MB) stays on the box; only summaries go to gpurun_out/
mkdir -p $R
part=${1:-all}
python3 -c "from meepoembedding_amd import _lib; import json; print(json.dumps(_lib.device_calibration(0)))" 2>/dev/null | tail -1 > $O/calibration_part$part.json
if [ "$part" = 1 ] || [ "$part" = all ]; then
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_find -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streams > $O/bench_profiled.json 2> $O/bench_profiled.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/pmc_fetch -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streams > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/pmc_write -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streams > /dev/null 2> $O/pmc_write.err
python3 tools/pmc_traffic.py "find_kernel<16, 2" $R/pmc_fetch $R/pmc_write --out $O/find_traffic.json --meta batch=262144 dim=64 keys=100000000 > /dev/null
echo "find traffic done"
timeout -k 10 300 python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
timeout -k 10 300 python3 bench.py --mode train --dist zipf --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_train_zipf.json 2> $O/bench_train_zipf.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_train -o run -- python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-streams > $O/bench_train_profiled.json 2> $O/bench_train_profiled.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/pmc_train_fetch -o run -- python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-streams > /dev/null 2> $O/pmc_train_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/pmc_train_write -o run -- python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-streams > /dev/null 2> $O/pmc_train_write.err
python3 - <<PY
import json, subprocess, sys
ks = {}
for k in ("find_prepare_kernel<16, 2, 64>", "bkt_apply_kernel<1, 16, true, false, false>"):
    r = subprocess.run([sys.executable, "tools/pmc_traffic.py", k, "$R/pmc_train_fetch", "$R/pmc_train_write"], capture_output=True, text=True)
    if r.returncode == 0:
        ks[k] = json.loads(r.stdout)
    else:
        print(k, r.stderr[-300:])
json.dump({"batch": 262144, "dim": 64, "keys": 100000000, "kernels": ks}, open("$O/step_traffic.json", "w"), indent=1)
PY
cp $(find $R/prof_find -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find $R/prof_train -name "*kernel_stats.csv" | head -1) $O/bench_train_kernel_stats.csv
echo "part 1 done"
fi
if [ "$part" = 2 ] || [ "$part" = all ]; then
for d in uniform zipf; do timeout -k 10 200 python3 tools/apply_trace.py 100000000 $d 2>&1 | grep apply_path >> $O/apply.txt; done
timeout -k 10 200 python3 tools/apply_trace.py 100000000 uniform - adam 2>&1 | grep apply_path >> $O/apply.txt
timeout -k 10 200 python3 tools/apply_trace.py 100000000 zipf - adam 2>&1 | grep apply_path >> $O/apply.txt
for d in uniform zipf; do
  rm -rf $R/at_$d && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/at_$d -o x -- python3 tools/apply_trace.py 100000000 $d > /dev/null 2>&1
  echo "== apply_trace.py $d: kernels" >> $O/apply.txt
  python3 tools/kernel_stats.py $R/at_$d bkt_ find_prepare "find_kernel<16, 2, 64>" >> $O/apply.txt
done
echo "apply traces done"
MEE_PMC_OUT=$R/pmc_sq bash tools/pmc_sq.sh
python3 tools/pmc_sq_summary.py $R/pmc_sq > $O/apply_sq.md
echo "sq done"
timeout -k 10 300 python3 tools/first_skewed_batch.py > $O/first_skewed.txt 2>&1
bash tools/dedup_kernels.sh $O/dedup.txt > /dev/null
echo "part 2 done"
fi
if [ "$part" = 3 ] || [ "$part" = all ]; then
MEE_PMC_OUT=$R/pmc_all bash tools/pmc_all.sh
python3 tools/kernel_rooflines.py $R/pmc_all > $O/kernel_rooflines.md
bash tools/world1_zipf.sh $O/world1 > $O/world1.txt 2>&1
echo "part 3 done"
fi
echo done > $O/done_$part

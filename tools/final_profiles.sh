# end-of-round evidence run on the GPU box: bench lines + rocprofv3 kernel stats + apply timings -> gpurun_out/final/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final
O=gpurun_out/final
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_find -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streams > $O/bench_profiled.json 2> $O/bench_profiled.err
timeout -k 10 300 python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-streams > $O/bench_train.json 2> $O/bench_train.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o run -- python3 bench.py --mode train --steps 100 --warmup 10 --no-cpu-baseline --no-streams > $O/bench_train_profiled.json 2> $O/bench_train_profiled.err
for d in uniform zipf; do timeout -k 10 200 python3 tools/tune_apply.py --dist $d --rounds 3 2>&1 | grep -E "^adagrad|round 1" >> $O/apply.txt; done
timeout -k 10 200 python3 tools/tune_apply.py --opt adam --rounds 3 2>&1 | grep -E "^adam|round 1" >> $O/apply.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zipf -o run -- python3 tools/apply_trace.py 100000000 zipf > /dev/null 2>&1
echo done > $O/done

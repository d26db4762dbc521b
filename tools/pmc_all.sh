# three rocprofv3 passes over tools/all_ops.py: kernel durations, FETCH_SIZE, WRITE_SIZE  -> gpurun_out/pmc_all/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_all
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_all/time -o run -- python3 tools/all_ops.py > gpurun_out/pmc_all/time.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_all/fetch -o run -- python3 tools/all_ops.py > gpurun_out/pmc_all/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_all/write -o run -- python3 tools/all_ops.py > gpurun_out/pmc_all/write.log 2>&1
echo done > gpurun_out/pmc_all/done

# three rocprofv3 passes over tools/all_ops.py: kernel durations, FETCH_SIZE, WRITE_SIZE  -> $P/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${MEE_PMC_OUT:-gpurun_out/pmc_all}   # raw counter CSVs are large: point MEE_PMC_OUT outside gpurun_out/ when only the summary is wanted
mkdir -p $P
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/time -o run -- python3 tools/all_ops.py > $P/time.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -o run -- python3 tools/all_ops.py > $P/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -o run -- python3 tools/all_ops.py > $P/write.log 2>&1
echo done > $P/done

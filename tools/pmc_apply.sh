# HBM traffic of the apply path's kernels: separate FETCH_SIZE / WRITE_SIZE passes over tools/apply_trace.py (uniform 256K-key batches)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_apply
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_apply/fetch -o run -- python3 tools/apply_trace.py 100000000 ${1:-uniform} > gpurun_out/pmc_apply/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_apply/write -o run -- python3 tools/apply_trace.py 100000000 ${1:-uniform} > gpurun_out/pmc_apply/write.log 2>&1
for k in "group_kernel<2" "apply_main_kernel<1, 16, 1, true>" "apply_main_kernel<1, 16, 1, false>" apply_dups_kernel apply_filed_kernel apply_big_kernel "find_kernel<16, 2, 64>"; do
  python3 tools/pmc_traffic.py "$k" gpurun_out/pmc_apply/fetch gpurun_out/pmc_apply/write > "gpurun_out/pmc_apply/$(echo $k | tr -c 'a-zA-Z0-9_\n' '_').json" || true
done

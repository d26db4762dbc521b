# SQ counters of the apply kernels on a uniform and a Zipf(1.05) stream
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_sq
for d in uniform zipf; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq/$d -o run -- python3 tools/apply_trace.py 100000000 $d > gpurun_out/pmc_sq/$d.log 2>&1
done

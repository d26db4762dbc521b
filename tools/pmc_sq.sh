# SQ counters of the apply path's kernels (VERDICT r2 #2c): one rocprofv3 --pmc pass per key stream over tools/apply_trace.py, both apply
# paths (1 = bucketed, 0 = group table).  -> gpurun_out/pmc_sq/<stream>/ ; summarise with tools/pmc_sq_summary.py -> profiles/r03_apply_sq.md
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_sq
for d in uniform zipf; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq/$d -o run -- python3 tools/apply_trace.py 100000000 $d 1,0 > gpurun_out/pmc_sq/$d.log 2>&1
done
echo done > gpurun_out/pmc_sq/done

# SQ counters of the apply path's kernels: one rocprofv3 --pmc pass per key stream over tools/apply_trace.py.
# -> $P/<stream>/ ; summarise with tools/pmc_sq_summary.py -> profiles/r04_apply_sq.md
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${MEE_PMC_OUT:-gpurun_out/pmc_sq}   # raw counter CSVs are large: point MEE_PMC_OUT outside gpurun_out/ when only the summary is wanted
mkdir -p $P
for d in uniform zipf; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/$d -o run -- python3 tools/apply_trace.py 100000000 $d > $P/$d.log 2>&1
done
echo done > $P/done

#!/bin/bash
# VERDICT r03 task 3: which shared resource refuses the empty vector-issue slots.  Two PMC passes over the default bench
# (each --kernel-trace + --pmc only): LDS side and vector-memory side, per kernel; then the concurrency of the wide kernels from
# a plain kernel trace (tools/concurrency.py).  -> gpurun_out/<tag>/issue_gaps.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04_issue_gaps}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
{
echo "# PMC pass 1 (LDS): per kernel, averages per launch"
bash tools/pmc_any.sh ${TAG}_lds "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" --no-host-in --no-workloads
echo
echo "# PMC pass 2 (vector memory / VALU): per kernel, averages per launch"
bash tools/pmc_any.sh ${TAG}_vmem "SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU" --no-host-in --no-workloads
echo
echo "# PMC pass 3 (texture addresser / data return)"
bash tools/pmc_any.sh ${TAG}_ta "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" --no-host-in --no-workloads
echo
echo "# which wide kernels are resident together (kernel trace of the default bench, middle of the timed region)"
bash tools/trace_concurrency.sh ${TAG}_conc
} > $OUT/issue_gaps.txt 2>&1
cat $OUT/issue_gaps.txt

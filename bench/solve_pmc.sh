#!/bin/bash
# usage (GPU box): bench/solve_pmc.sh [B] [counter sets...]   -- SQ / LDS / I-cache counters of k_al_ilqr on B random landing problems (N = 40)
B=${1:-4096}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/solve_pmc
mkdir -p $OUT
cd $R
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE")
if [ -n "$QLN_PMC_ONLY" ]; then SETS=("${SETS[@]:$QLN_PMC_ONLY}"); fi
i=0
for C in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench/solve_sweep.py $B 40 14 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 $R/bench/pmc_summary.py k_al_ilqr $OUT/p* | tee $OUT/summary.txt

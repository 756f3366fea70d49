#!/bin/bash
# usage: bench/pmc.sh TAG "<bench args>" "CTRS pass1" "CTRS pass2" ...   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass per counter set (never combined with tracing domains other than kernel-trace).
TAG=$1; shift; ARGS=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
cd /tmp
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_p$i -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-other > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_${TAG}_p$i.log; exit 1; }
done
python3 $R/bench/pmc_summary.py k_constraint_jacobian $R/gpurun_out/pmc_${TAG}_p* | tee $R/gpurun_out/pmc_${TAG}_summary.txt

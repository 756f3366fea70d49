#!/bin/bash
# usage (GPU box): bench/solve_evidence.sh   -- the records under profiles/ that DESIGN.md 4.6 quotes for qln_solve
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/solve_evidence
mkdir -p $OUT
cd $R
{
  for B in 1024 16384 65536; do python3 bench/solve_sweep.py $B 40 14 full 2>&1 | grep -v amdgpu.ids; done
} > $OUT/solve_sweep.txt
{
  echo "# qln_solve, N = 40, k_trans = 14, default options: the two register budgets of k_al_ilqr (tuning build, QLN_ILQR_OCC)"
  echo "# OCC=1: 512 registers, one wave per SIMD (4 per CU); OCC=2: 256 registers, two per SIMD (8 per CU at N = 40: 19.8 KB of LDS each)"
  for B in 1024 4096 16384 65536; do for O in 1 2; do
    echo -n "B=$B OCC=$O  "; QLN_LIB_PATH=quadruped_landing_amd/csrc/libqln_hip_tuning.so QLN_ILQR_OCC=$O python3 bench/solve_sweep.py $B 40 14 2>&1 | grep "status" | cut -c72-
  done; done
} > $OUT/solve_occupancy.txt
cat $OUT/solve_occupancy.txt

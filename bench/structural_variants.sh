#!/bin/bash
# A/B the tuning build's instantiations of the structural-format kernel on one box, interleaved:
#   bench/structural_variants.sh [workload] [rounds] [variants...]   (needs `make -C quadruped_landing_amd/csrc tuning`)
# variants of the tuning build (qln_kernels.hip, launch_constraint_jacobian): 0 = shipping, 11-15 chunk sizes / register budgets,
# 16 = two 20-knot sub-tiles + three waves per SIMD, 17 = the same with two waves, 18 = one image + two waves, 19 = three 14-knot sub-tiles
WL=${1:-config3}; R=${2:-2}; shift 2
VARS=${@:-0 16 17 18 19}
export QLN_LIB_PATH=$PWD/quadruped_landing_amd/csrc/libqln_hip_tuning.so
export QLN_ABLATE_PLACED=1
for i in $(seq $R); do
  for V in $VARS; do
    printf "variant %-3s " $V; QLN_VARIANT=$V python bench/ablate.py $WL structural 2>&1 | grep -E "fused c\+J  |J only|c only" | tr '\n' ' '; echo
  done
done

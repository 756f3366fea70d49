#!/bin/bash
# A/B the tuning build's instantiations of the structural-format kernel on one box, interleaved:
#   bench/structural_variants.sh [workload] [rounds]        (needs `make -C quadruped_landing_amd/csrc tuning`)
WL=${1:-config3}; R=${2:-2}
export QLN_LIB_PATH=$PWD/quadruped_landing_amd/csrc/libqln_hip_tuning.so
for i in $(seq $R); do
  for V in 0 11 12 13 14 15; do
    printf "variant %-3s " $V; QLN_VARIANT=$V python bench/ablate.py $WL structural 2>&1 | grep -E "fused c\+J  |J only|c only" | tr '\n' ' '; echo
  done
done

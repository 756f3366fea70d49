#!/bin/bash
# A/B two in-tree builds of the library on the SAME box, interleaved: bench/ab.sh libA.so libB.so [workload] [rounds]
A=$1; B=$2; WL=${3:-config3}; R=${4:-3}
for i in $(seq $R); do
  for L in $A $B; do
    printf "%-28s " $(basename $L); QLN_LIB_PATH=$PWD/quadruped_landing_amd/csrc/$L python bench/ablate.py $WL 2>&1 | grep -E "fused c\+J  " 
  done
done

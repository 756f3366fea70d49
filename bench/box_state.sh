#!/bin/bash
# What state is this box in?  Clocks / power / partition modes next to the fused-launch and memset times, to relate
# the "slow" and "fast" boxes to something observable.   bench/box_state.sh
rocm-smi --showclocks --showpower --showmaxpower --showmemuse --showcomputepartition --showmemorypartition --showperflevel 2>&1 | grep -v "^=\|^$" | head -40
python bench/ablate.py config3 2>&1 | grep -E "fused c\+J  |memset"
rocm-smi --showclocks 2>&1 | grep -E "mclk|sclk|fclk|socclk" | head -8

#!/bin/bash
# same-box A/B of the 3x3 layers: conv_pc_kernel (CHORE_CONV_MW=0) against conv_mw_kernel (CHORE_CONV_MW=all), alternating runs
#   usage (on the GPU box): scripts/conv_mw_ab.sh [rounds] > gpurun_out/r6_conv_mw_ab.txt
rounds=${1:-2}
for r in $(seq $rounds); do
  for m in 0 all; do
    echo "CHORE_CONV_MW=$m run $r: $(CHORE_CONV_MW=$m python scripts/conv_layer_ab.py fp16x3 3x3 2>&1 | tail -1)"
  done
done

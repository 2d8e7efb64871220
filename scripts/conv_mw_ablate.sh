#!/bin/bash
# phase breakdown of conv_mw_kernel by compile-time ablation (conv_mw.hip MW_DBG; libraries from scripts/build_variant.sh mwN):
#   usage (GPU box): scripts/conv_mw_ablate.sh "256->128 @128" 0 1 2 3 4 8 16 32 7 12 19
layer=$1; shift
for v in "$@"; do
  lib=chore_amd/csrc/build_ab/libchore_hip_mw$v.so
  [ "$v" = 0 ] && lib=chore_amd/csrc/libchore_hip.so
  echo "MW_DBG=$v: $(CHORE_HIP_LIB=$lib CHORE_CONV_MW=all CONV_AB_ONLY="$layer" python scripts/conv_layer_ab.py fp16x3 3x3 2>&1 | tail -1)"
done

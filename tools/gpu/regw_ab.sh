#!/bin/bash
# A/B of conv1x1_regw_kernel (filters resident in registers) against the tile / streaming kernels on the Mixed_5 / Mixed_6 block-entry shapes,
# through the C ABI (tools/conv_bench.py).
#   usage (on the GPU box): tools/gpu/regw_ab.sh [layer:which ...] > gpurun_out/regw_ab.txt
#   RELU=' ' for dense random operands (default: half zeros, as behind a ReLU)
cd "$(dirname "$0")/../.."
export DIN_OPTIONS_FROM_ENV=1
specs=("$@")
[ ${#specs[@]} -eq 0 ] && specs=(inc_6e_1x1_768:fwd inc_6e_entry_576:fwd inc_6b_entry_448:fwd k_1x1_768_768:fwd k_1x1_768_768:dgrad inc_6e_1x1_768_b4:fwd k_1x1_768_768_b4:dgrad \
                                 inc_5b_entry_176:fwd inc_5c_entry_176:fwd inc_5d_entry_176:fwd inc_5c_pool_64:fwd k_1x1_256_240:dgrad)
for spec in "${specs[@]}"; do
  l=${spec%%:*}; w=${spec##*:}
  for m in 0 2; do
    echo "== $l $w DIN_CONV_REGW=$m"
    DIN_CONV_REGW=$m timeout 300 python tools/conv_bench.py --layer $l --which $w --iters 200 ${RELU:---relu} 2>&1 | tail -1
  done
done

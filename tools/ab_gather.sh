#!/bin/bash
# A/B of two builds of the library on the gather-family layers (tools/conv_bench.py): $1 = baseline .so (DIN_LIB_PATH), the in-tree build is B.
# Interleaved A B A B per layer so that box drift shows.
BASE=${1:-knock_build/libdin_hip_base.so}
LAYERS=${LAYERS:-"inc_6e_7x1 inc_6e_1x7 inc_6e_1x1_768 k_1x1_192 inc_6a_3x3 inc_6c_1x7 inc_5b_5x5 inc_4a_3x3 inc_6a_dbl3"}
for l in $LAYERS; do
  for w in fwd dgrad; do
    for r in 1 2; do
      echo -n "A "; DIN_LIB_PATH=$BASE python tools/conv_bench.py --layer $l --which $w --iters 30 | tail -1
      echo -n "B "; python tools/conv_bench.py --layer $l --which $w --iters 30 | tail -1
    done
  done
done

#!/bin/bash
# A/B of the per-lane k-walk (conv_gather_fast_kernel<..., LANEK>) against the general loop: layer microbenchmarks, then the whole step.
#   usage: tools/ab_lanek.sh <out dir>
OUT=${1:-gpurun_out/lanek}
mkdir -p $OUT
for L in inc_4a_3x3 inc_6c_1x7 inc_6c_7x1 inc_6c_7x1_192; do
  for W in fwd dgrad; do
    for M in 1 0; do
      DIN_CONV_LANEK=$M python tools/conv_bench.py --layer $L --which $W --relu --iters 30 2>&1 | tail -1 | sed "s/^/LANEK=$M /"
    done
  done
done > $OUT/layers.txt 2>&1
cat $OUT/layers.txt
for i in 1 2; do
  for M in 1 0; do
    DIN_OPTIONS_FROM_ENV=1 DIN_CONV_LANEK=$M python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/b32_lanek${M}_$i.log 2>&1
    DIN_OPTIONS_FROM_ENV=1 DIN_CONV_LANEK=$M python bench.py --global-batch 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/b4_lanek${M}_$i.log 2>&1
  done
done
for f in $OUT/b*.log; do echo -n "$f "; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done

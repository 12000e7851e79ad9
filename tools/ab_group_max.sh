#!/bin/bash
# A/B of the layers per grouped weight-gradient launch (DIN_GROUP_WGRAD_MAX) on the 4-clip and the 32-clip step, alternating runs on one box.
#   usage: tools/ab_group_max.sh <out dir> "<sizes for 32 clips>" "<sizes for 4 clips>"
OUT=${1:-gpurun_out/groupmax}
G32=${2:-"2 4 8"}
G4=${3:-"8 12"}
mkdir -p $OUT
for i in 1 2; do
  for G in $G32; do
    DIN_OPTIONS_FROM_ENV=1 DIN_GROUP_WGRAD_MAX=$G python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/b32_g${G}_$i.log 2>&1
  done
  DIN_OPTIONS_FROM_ENV=1 DIN_GROUP_WGRAD=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/b32_g0_$i.log 2>&1
  for G in $G4; do
    DIN_OPTIONS_FROM_ENV=1 DIN_GROUP_WGRAD_MAX=$G python bench.py --global-batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/b4_g${G}_$i.log 2>&1
  done
done
for f in $OUT/b*.log; do echo -n "$f "; grep '"metric"' $f | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; done

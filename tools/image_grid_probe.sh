#!/bin/bash
# image-layer forward kernel (uint8 loader) at several persistent-grid sizes (rocprofv3 kernel stats of 3 bench steps)
cd /tmp && export TMPDIR=/tmp
for g in 512 768 1024; do
  DIN_CONV_IMAGE_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ig_$g -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== DIN_CONV_IMAGE_GRID=$g"
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/ig_$g -name "*kernel_stats.csv" | head -1)
  grep -E "conv_small_kernel<1" $f | cut -d, -f1-4 | cut -c1-160
done

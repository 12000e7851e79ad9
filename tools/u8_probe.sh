#!/bin/bash
# per-kernel time of the image layer with and without the fused uint8 loader (rocprofv3 kernel stats of 3 bench steps)
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  DIN_CONV_U8=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/u8_$m -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== DIN_CONV_U8=$m"
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/u8_$m -name "*kernel_stats.csv" | head -1)
  grep -E "conv_small_kernel<1|conv_wgrad_small_kernel<1|prep_nhwc" $f | cut -d, -f1-4 | cut -c1-160
done

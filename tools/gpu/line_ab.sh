#!/bin/bash
# A/B of the tap-line kernel (tools/probes/line_probe, standalone) against the shipped gather kernels (tools/conv_bench.py) on one box.
#   usage: tools/gpu/line_ab.sh <tag> [reps]
set -u
cd "$(dirname "$0")/../.."
TAG=$1; REPS=${2:-3000}
OUT=gpurun_out; mkdir -p $OUT
timeout 300 tools/probes/line_probe $REPS > $OUT/${TAG}_line_probe.txt 2>&1; echo "probe rc=$?"
cat $OUT/${TAG}_line_probe.txt
if [ "${SKIP_BENCH:-0}" != 1 ]; then
  for spec in "inc_6e_7x1 fwd" "inc_6e_1x7 fwd" "inc_6e_7x1 dgrad" "inc_6c_7x1_192 fwd" "inc_6b_7x1 fwd" "inc_6e_1x1_768 fwd"; do
    set -- $spec
    timeout 300 python tools/conv_bench.py --layer $1 --which $2 --iters $REPS 2>&1 | tail -1
  done | tee $OUT/${TAG}_shipped.txt
fi

# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
X=knock_build/experiments/libdin_hip.so
for l in inc_6c_1x7 inc_6c_7x1 inc_6b_1x7 inc_6b_7x1 inc_6c_7x1_192; do
  for w in fwd dgrad; do
    for r in 1 2; do
      echo -n "128-px 8w   "; python tools/conv_bench.py --layer $l --which $w --iters 2000 | tail -1
      echo -n "256-px 16w  "; DIN_LIB_PATH=$X DIN_CONV_TILE=256 DIN_CONV_W16=1 python tools/conv_bench.py --layer $l --which $w --iters 2000 | tail -1
    done
  done
done

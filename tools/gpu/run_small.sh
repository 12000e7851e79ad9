# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
X=knock_build/experiments/libdin_hip.so
for l in inc_6e_7x1 inc_6e_1x7 inc_6c_1x7 inc_6b_1x7 k_1x1_192 inc_6c_7x1_192; do
  for w in fwd dgrad; do
    for r in 1 2; do
      echo -n "2 WG x 8 waves, 64-deep  "; DIN_LIB_PATH=$X python tools/conv_bench.py --layer $l --which $w --iters 2000 | tail -1
      echo -n "3 WG x 4 waves, 32-deep  "; DIN_LIB_PATH=$X DIN_CONV_WG3=1 python tools/conv_bench.py --layer $l --which $w --iters 2000 | tail -1
    done
  done
done

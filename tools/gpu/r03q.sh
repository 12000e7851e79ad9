#!/bin/bash
# steady-state (power-capped) comparison of the 128x192 two-workgroup tile against the 256x192 sixteen-wave tile (a fifth fewer staged bytes per FLOP)
mkdir -p gpurun_out/r03q
P=tools/clock_probe.sh
for w in fwd dgrad; do
  $P gpurun_out/r03q/t128_$w.csv -- python tools/conv_bench.py --layer inc_6e_7x1 --which $w --iters 12000
  DIN_CONV_TILE=256 DIN_CONV_W16=1 $P gpurun_out/r03q/t256_$w.csv -- python tools/conv_bench.py --layer inc_6e_7x1 --which $w --iters 12000
done
DIN_CONV_WAVEGRID=24 $P gpurun_out/r03q/g24_fwd.csv -- python tools/conv_bench.py --layer inc_6e_7x1 --which fwd --iters 12000

#!/bin/bash
# s_setprio 1 over the MFMA stream of the interleaved FASTK k-step (knock_build/prio) vs the shipped build; steady-state clocks
mkdir -p gpurun_out/r03r
for l in inc_6e_7x1 inc_6e_1x1_768; do
for w in fwd dgrad; do
  for r in 1 2; do
    echo -n "ship "; python tools/conv_bench.py --layer $l --which $w --iters 10000 | tail -1
    echo -n "prio "; DIN_LIB_PATH=$PWD/knock_build/prio/libdin_hip.so python tools/conv_bench.py --layer $l --which $w --iters 10000 | tail -1
  done
done
done > gpurun_out/r03r/ab.log 2>&1
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03r/bench_ship_$r.json
  DIN_LIB_PATH=$PWD/knock_build/prio/libdin_hip.so python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03r/bench_prio_$r.json
done

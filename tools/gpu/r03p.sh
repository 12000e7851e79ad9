#!/bin/bash
# DIN_CONV_RING=3 (three 32-deep stages) vs the shipped two 64-deep stages on the general-loop 8-wave tiles; steady-state clocks (--iters in the thousands)
mkdir -p gpurun_out/r03p
DIN_CONV_RING=3 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv" 2>&1 | tail -3 > gpurun_out/r03p/tests_ring3.log
for l in inc_4a_3x3:2000 inc_6c_1x7:8000 inc_6a_3x3:3000; do
  L=${l%%:*}; N=${l##*:}
  for w in fwd dgrad; do
    for r in 1 2; do
      echo -n "ring2 "; python tools/conv_bench.py --layer $L --which $w --iters $N | tail -1
      echo -n "ring3 "; DIN_CONV_RING=3 python tools/conv_bench.py --layer $L --which $w --iters $N | tail -1
    done
  done
done > gpurun_out/r03p/ab.log 2>&1
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03p/bench_ring2_$r.json
  DIN_CONV_RING=3 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03p/bench_ring3_$r.json
done

#!/bin/bash
# batch-statistics BatchNorm: cap on the statistics kernels' workgroup count (their closing fp64 atomics serialise in L2)
mkdir -p gpurun_out/r03t
python -m pytest tests -m gpu -x -q -k "bn or batch_stat or BatchNorm or batchnorm" 2>&1 | tail -2 > gpurun_out/r03t/tests.log
for cap in 0 1024 256 512 2048 0 1024; do
  DIN_BN_MAX_BLOCKS=$cap python bench.py --bn-mode batch --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03t/bench_cap${cap}_$RANDOM.json
done

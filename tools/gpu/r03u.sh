#!/bin/bash
# batch-statistics BatchNorm: row-walk apply kernels (per-channel constants hoisted, four rows in flight) vs the element-per-iteration kernels
mkdir -p gpurun_out/r03u
python -m pytest tests -m gpu -x -q -k "bn or batch_stat or BatchNorm or batchnorm" 2>&1 | tail -2 > gpurun_out/r03u/tests.log
for r in 1 2; do
  DIN_BN_APPLY_ROWS=0 python bench.py --bn-mode batch --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03u/bench_old_$r.json
  python bench.py --bn-mode batch --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03u/bench_rows_$r.json
done
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03u/p -o x -- python $R/bench.py --bn-mode batch --no-cpu-baseline --no-extras > $R/gpurun_out/r03u/run.log 2>&1)
find gpurun_out/r03u/p -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03u/kernel_stats.csv \; ; rm -rf gpurun_out/r03u/p

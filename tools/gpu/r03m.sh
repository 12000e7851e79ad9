#!/bin/bash
mkdir -p gpurun_out/r03m; cd /root/repo; rm -f gpurun_out/r03m/ab.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "stem or image" 2>&1 | tail -3 >> gpurun_out/r03m/ab.log
for i in 1 2 3; do for m in 1 2; do
  DIN_CONV_SMALL_NBUF=$m python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nbuf=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03m/ab.log
done; done
cat gpurun_out/r03m/ab.log

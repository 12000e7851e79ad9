#!/bin/bash
mkdir -p gpurun_out/r03k; cd /root/repo; rm -f gpurun_out/r03k/ab.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_din_model.py -q -x 2>&1 | tail -5 > gpurun_out/r03k/tests.log
for i in 1 2; do for m in 0 1; do
  DIN_CONV_STREAM=$m python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('stream=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03k/ab.log
done; done
cat gpurun_out/r03k/tests.log gpurun_out/r03k/ab.log

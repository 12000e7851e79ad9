#!/bin/bash
mkdir -p gpurun_out/r03g; cd /root/repo; rm -f gpurun_out/r03g/ab3.log
python -m pytest tests/test_gpu_kernels.py -q -x -k "maxpool or pools" 2>&1 | tail -3 > gpurun_out/r03g/tests.log
python tools/pool_bench.py 2>&1 | grep maxpool > gpurun_out/r03g/pool_bench5.log
for i in 1 2; do for m in 0 1; do
  DIN_MAXPOOL_ROWS=$m python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('rows=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03g/ab3.log
done; done
cat gpurun_out/r03g/tests.log gpurun_out/r03g/pool_bench5.log gpurun_out/r03g/ab3.log

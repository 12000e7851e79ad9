#!/bin/bash
mkdir -p gpurun_out/r03m; cd /root/repo; rm -f gpurun_out/r03m/ab.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "halo or big_tile or 1x7 or 7x1" 2>&1 | tail -3 >> gpurun_out/r03m/ab.log
for i in 1 2 3; do for m in old new; do
  if [ $m = old ]; then export DIN_LIB_PATH=/root/repo/knock_build/head/libdin_hip.so; else unset DIN_LIB_PATH; fi
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03m/ab.log
done; done
unset DIN_LIB_PATH
timeout 1200 python -m pytest tests/test_gpu_din_model.py -q -x 2>&1 | tail -3 >> gpurun_out/r03m/ab.log
cat gpurun_out/r03m/ab.log

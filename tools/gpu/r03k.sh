#!/bin/bash
mkdir -p gpurun_out/r03k; cd /root/repo; rm -f gpurun_out/r03k/ab3.log
for i in 1 2 3; do for m in 262144 32768; do
  DIN_CONV_STREAM_MINPIX=$m python bench.py --global-batch 4 --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b4 minpix=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03k/ab3.log
done; done
cat gpurun_out/r03k/ab3.log

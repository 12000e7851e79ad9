#!/bin/bash
mkdir -p gpurun_out/r03k; cd /root/repo; rm -f gpurun_out/r03k/ab2.log
for i in 1 2 3; do for m in 0 1; do
  DIN_CONV_STREAM_WIDE=$m python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('wide=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03k/ab2.log
done; done
cat gpurun_out/r03k/ab2.log

#!/bin/bash
# row max-pool kernels: step A/B
mkdir -p gpurun_out/r03g; cd /root/repo; rm -f gpurun_out/r03g/ab2.log
for i in 1 2 3; do for m in 0 2; do
  DIN_MAXPOOL_ROWS=$m python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('rows=$m', j['value'], j['ms_per_step'])" >> gpurun_out/r03g/ab2.log
done; done
cat gpurun_out/r03g/ab2.log

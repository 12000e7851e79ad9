#!/bin/bash
mkdir -p gpurun_out/r03l; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "pool" 2>&1 | tail -3 > gpurun_out/r03l/tests.log
python tools/pool_bench.py 2>&1 | grep avgpool > gpurun_out/r03l/avgpool.log
cat gpurun_out/r03l/tests.log gpurun_out/r03l/avgpool.log

#!/bin/bash
mkdir -p gpurun_out/r03i; cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03i/gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/r03i/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03i/smoke.log
tail -8 gpurun_out/r03i/gpu_tests.log; tail -3 gpurun_out/r03i/smoke.log

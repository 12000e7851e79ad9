#!/bin/bash
mkdir -p gpurun_out/r03j; cd /root/repo
DIN_CONV_STREAM_WAVES=16 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "two_destinations or dgrad_multi_source or (test_conv_fwd_dgrad_wgrad and bf16)" 2>&1 | tail -5 > gpurun_out/r03j/tests16.log
echo "== 8 waves" > gpurun_out/r03j/bench5.log
timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03j/bench5.log
echo "== 16 waves" >> gpurun_out/r03j/bench5.log
DIN_CONV_STREAM_WAVES=16 timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03j/bench5.log
cat gpurun_out/r03j/tests16.log gpurun_out/r03j/bench5.log

#!/bin/bash
mkdir -p gpurun_out/r03j; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "two_destinations or dgrad_multi_source or (test_conv_fwd_dgrad_wgrad and bf16)" 2>&1 | tail -5 > gpurun_out/r03j/tests.log
echo "== BN by cout" > gpurun_out/r03j/bench4.log
timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03j/bench4.log
echo "== BN=96 forced" >> gpurun_out/r03j/bench4.log
DIN_CONV_STREAM_BN=96 timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03j/bench4.log
cat gpurun_out/r03j/tests.log gpurun_out/r03j/bench4.log

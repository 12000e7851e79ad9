#!/bin/bash
mkdir -p gpurun_out/r03j; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "two_destinations or dgrad_multi_source or (test_conv_fwd_dgrad_wgrad and bf16 and st_)" 2>&1 | tail -5 > gpurun_out/r03j/tests.log
timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03j/bench3.log
echo "== knock 7" >> gpurun_out/r03j/bench3.log
DIN_GATHER_KNOCK=7 timeout 600 python tools/conv_stream_bench.py 2>&1 | grep -v amdgpu.ids | head -7 >> gpurun_out/r03j/bench3.log
cat gpurun_out/r03j/tests.log gpurun_out/r03j/bench3.log

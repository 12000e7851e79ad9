#!/bin/bash
# late round 3: SQ counters of the FASTK gather kernel before / after the in-wave interleaved k-step, and sclk / power under more kernels
mkdir -p gpurun_out/r03o
tools/pmc_layer.sh inc_6e_7x1 fwd conv_gather_fast_kernel gpurun_out/r03o/pmc_ilv > /dev/null 2>&1
tools/pmc_layer.sh inc_6e_7x1 fwd conv_gather_fast_kernel gpurun_out/r03o/pmc_base DIN_LIB_PATH=knock_build/libdin_hip_base.so > /dev/null 2>&1
rm -rf gpurun_out/r03o/pmc_*/p[0-9]
P=tools/clock_probe.sh
$P gpurun_out/r03o/halo.csv -- python tools/conv_bench.py --layer inc_5d_3x3 --which fwd --iters 12000
$P gpurun_out/r03o/c4a.csv -- python tools/conv_bench.py --layer inc_4a_3x3 --which fwd --iters 2500
$P gpurun_out/r03o/c4a_dgrad.csv -- python tools/conv_bench.py --layer inc_4a_3x3 --which dgrad --iters 2500
$P gpurun_out/r03o/s3b.csv -- python tools/conv_bench.py --layer inc_3b_1x1 --which fwd --iters 12000
$P gpurun_out/r03o/w4a.csv -- python tools/conv_bench.py --layer inc_4a_3x3 --which wgrad --iters 3000
$P gpurun_out/r03o/w5b.csv -- python tools/conv_bench.py --layer inc_5b_5x5 --which wgrad --iters 12000
$P gpurun_out/r03o/pool.csv -- env DIN_POOL_ITERS=1500 python tools/pool_bench.py
$P gpurun_out/r03o/b4.csv -- python bench.py --global-batch 4 --no-cpu-baseline --no-extras --steps 600 --warmup 5

#!/bin/bash
mkdir -p gpurun_out/r03h; cd /root/repo
DIN_BENCH_TORCH_PROFILE=gpurun_out/r03h/torch_b32.txt python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 3 > gpurun_out/r03h/b32.log 2>&1
sed -n '/==== ATen/,$p' gpurun_out/r03h/torch_b32.txt | head -80

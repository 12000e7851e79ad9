#!/bin/bash
mkdir -p gpurun_out/r03h; cd /root/repo
DIN_BENCH_TORCH_PROFILE=gpurun_out/r03h/torch_b4.txt python bench.py --global-batch 4 --force-buckets --no-cpu-baseline --no-extras --steps 5 --warmup 3 > gpurun_out/r03h/b4.log 2>&1
tail -2 gpurun_out/r03h/b4.log

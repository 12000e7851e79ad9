DIN_BENCH_TORCH_PROFILE=gpurun_out/torch_prof_b4.txt timeout 600 python bench.py --steps 2 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120

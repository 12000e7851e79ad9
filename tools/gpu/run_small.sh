timeout 600 python bench.py --steps 5 --warmup 2 --forward-only --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_fwd_only.json | cut -c1-420

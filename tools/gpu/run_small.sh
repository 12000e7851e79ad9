timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --host-images 2>&1 | tail -1 | tee gpurun_out/bench_host_images.json | cut -c1-330

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "roi" 2>&1 | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140

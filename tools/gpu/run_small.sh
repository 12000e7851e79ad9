timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "roi" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200

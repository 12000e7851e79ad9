timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180

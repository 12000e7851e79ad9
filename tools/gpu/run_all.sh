timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_last.json

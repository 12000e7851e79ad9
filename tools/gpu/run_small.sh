# scratch script for one-off gpurun calls during tuning (see the other scripts in this directory for the repeatable ones)
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300

for w in 1024 512; do
echo blocks=$w
DIN_WGRAD_BLOCKS=$w timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
DIN_WGRAD_BLOCKS=$w timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done

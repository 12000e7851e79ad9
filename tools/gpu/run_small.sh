for v in 0 96 64; do
DIN_WGRAD_SMALLM=$v timeout 600 python bench.py --steps 20 --warmup 3 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done
for v in 0 96; do
DIN_WGRAD_SMALLM=$v timeout 600 python bench.py --steps 20 --warmup 3 --global-batch 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done

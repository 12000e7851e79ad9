timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for r in 1 0; do
echo reduce_stream=$r
DIN_REDUCE_STREAM=$r timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
DIN_REDUCE_STREAM=$r timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done

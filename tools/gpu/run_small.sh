timeout 1500 python -m pytest tests/test_gpu_din_model.py -x -q 2>&1 | tail -4
for f in 1 0; do
echo fuse_wgrad=$f
DIN_FUSE_WGRAD=$f timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
DIN_FUSE_WGRAD=$f timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "two_destinations" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_din_model.py -x -q 2>&1 | tail -4
for f in 1 0; do
echo fuse_pool=$f
DIN_FUSE_POOL=$f timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
DIN_FUSE_POOL=$f timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -2
python tools/conv_bench.py --layer inc_6a_3x3 --which dgrad --iters 10 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180

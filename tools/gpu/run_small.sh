python tools/conv_bench.py --layer inc_3b_1x1_80 --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer inc_4a_3x3 --which fwd --iters 10 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer inc_5b_1x1 --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer inc_6e_7x1 --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180

timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "halo and bf16" 2>&1 | tail -15

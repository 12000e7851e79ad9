timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cin288" 2>&1 | tail -2

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tile or multi_source" 2>&1 | tail -5

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "second_stream or variant" 2>&1 | tail -8

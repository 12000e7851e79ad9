timeout 900 python -m pytest tests/test_gpu_din_model.py tests/test_gpu_kernels.py -q -m gpu -k "inception_bf16 or tile160" 2>&1 | tail -6

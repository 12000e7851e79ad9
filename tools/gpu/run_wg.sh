DIN_WGRAD_RING=3 timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_fwd_dgrad_wgrad and bf16 or bn_fold" 2>&1 | tail -4

timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fwd_dgrad_wgrad and bf16" 2>&1 | tail -3
for L in inc_6b_1x1 inc_6c_1x7 inc_5d_3x3 inc_4a_3x3 vgg_conv3_2 vgg_conv4_2; do
  for P in 0 1; do echo "ring=$P"; DIN_WGRAD_RING=$P timeout 300 python tools/conv_bench.py --layer $L --which wgrad 2>&1 | tail -1; done
done

timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "halo and bf16" 2>&1 | tail -4
for L in inc_4a_3x3 inc_5d_3x3 inc_6c_1x7; do
  for W in fwd dgrad; do
   for T in 0 1; do echo "halo=$T"; DIN_CONV_HALO=$T timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1; done
  done
done

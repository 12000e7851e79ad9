timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stem or 3x3_s1_p1 or big_tile" 2>&1 | tail -3
for L in inc_2a_3x3 inc_2b_3x3; do for W in fwd dgrad; do
  echo "== $L $W"; timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1
done; done

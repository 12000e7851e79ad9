for L in inc_4a_3x3 inc_6b_1x1 inc_6c_1x7; do
  for W in fwd dgrad; do
   for T in 0 8; do echo "pipe=$T"; DIN_CONV_HALO=0 DIN_CONV_PIPE=$T timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1; done
  done
done

for L in inc_5b_5x5 inc_3b_1x1; do
  for W in fwd dgrad; do
   for T in 0 128; do echo "tile=$T"; DIN_CONV_TILE=$T timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1; done
  done
done

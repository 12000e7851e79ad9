for L in inc_6c_1x7; do
  for W in fwd dgrad; do
   for T in 0 4; do echo "pipe=$T"; DIN_CONV_HALO=0 DIN_CONV_PIPE=$T timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1; done
  done
done
for T in 0 4; do echo "pipe=$T 96"; DIN_CONV_HALO=0 DIN_CONV_PIPE=$T timeout 300 python tools/conv_bench.py --layer inc_5d_3x3 --which fwd 2>&1 | tail -1; done
for T in 0 4; do echo "pipe=$T 3b fwd (80->96 tile)"; DIN_CONV_PIPE=$T timeout 300 python tools/conv_bench.py --layer inc_3b_1x1 --which fwd 2>&1 | tail -1; done

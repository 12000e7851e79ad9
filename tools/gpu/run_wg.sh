for L in inc_5b_1x1 inc_5d_3x3 inc_6b_1x1 inc_6c_1x7 inc_4a_3x3 inc_2b_3x3 vgg_conv3_2 vgg_conv4_2; do
  for W in fwd dgrad wgrad; do timeout 300 python tools/conv_bench.py --layer $L --which $W 2>&1 | tail -1; done
done

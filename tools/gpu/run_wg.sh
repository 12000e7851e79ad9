for L in inc_6c_1x7 inc_5d_3x3 vgg_conv3_2 vgg_conv4_2 vgg_conv2_2; do
   for T in 1 2; do echo "ring=$T"; DIN_WGRAD_RING=$T timeout 300 python tools/conv_bench.py --layer $L --which wgrad 2>&1 | tail -1; done
done

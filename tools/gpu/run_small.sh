for l in inc_6e_7x1 k_7x1_768 inc_6e_1x1_768 inc_4a_3x3; do
python tools/conv_bench.py --layer $l --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
DIN_CONV_TILE=256 DIN_CONV_PIPE=16 python tools/conv_bench.py --layer $l --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
DIN_CONV_TILE=256 python tools/conv_bench.py --layer $l --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
done

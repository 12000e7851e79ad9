# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "pf_1x1 or (test_conv_fwd_dgrad_wgrad and 1x1)" 2>&1 | tail -3
for l in inc_6e_1x1_768 k_1x1_1536 k_1x1_192 inc_5d_1x1_288; do
  for r in 1 2; do
    echo "touch  $(DIN_CONV_STREAM=0 python tools/conv_bench.py --layer $l --which fwd --iters 2000 | tail -1)"
    echo "plain  $(DIN_CONV_STREAM=0 DIN_CONV_L2_TOUCH=0 python tools/conv_bench.py --layer $l --which fwd --iters 2000 | tail -1)"
  done
done
timeout 600 python bench.py $B > $O/r04u_b32_touch.json 2> $O/r04u_b32_touch.err
DIN_CONV_L2_TOUCH=0 timeout 600 python bench.py $B > $O/r04u_b32_plain.json 2> $O/r04u_b32_plain.err
timeout 600 python bench.py $B > $O/r04u_b32_touch2.json 2> $O/r04u_b32_touch2.err
DIN_CONV_L2_TOUCH=0 timeout 600 python bench.py $B > $O/r04u_b32_plain2.json 2> $O/r04u_b32_plain2.err
python tools/bench_summary.py $O/r04u_*.json

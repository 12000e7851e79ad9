# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "wgrad or pipe" 2>&1 | tail -3
for l in inc_6e_7x1 inc_6c_1x7 inc_6b_1x7 inc_6c_7x1_192 inc_6a_3x3; do
  for r in 1 2; do
    echo -n "light (0.6)   "; python tools/conv_bench.py --layer $l --which wgrad --iters 1000 | tail -1
    echo -n "light (0.5)   "; DIN_WGRAD_LIGHT_COST=0.5 python tools/conv_bench.py --layer $l --which wgrad --iters 1000 | tail -1
    echo -n "light (0.75)  "; DIN_WGRAD_LIGHT_COST=0.75 python tools/conv_bench.py --layer $l --which wgrad --iters 1000 | tail -1
    echo -n "uniform       "; DIN_WGRAD_LIGHT=0 python tools/conv_bench.py --layer $l --which wgrad --iters 1000 | tail -1
  done
done
timeout 600 python bench.py $B > $O/r04l_b32_light.json 2> $O/r04l_b32_light.err
DIN_WGRAD_LIGHT=0 timeout 600 python bench.py $B > $O/r04l_b32_uniform.json 2> $O/r04l_b32_uniform.err
timeout 600 python bench.py $B > $O/r04l_b32_light2.json 2> $O/r04l_b32_light2.err
DIN_WGRAD_LIGHT=0 timeout 600 python bench.py $B > $O/r04l_b32_uniform2.json 2> $O/r04l_b32_uniform2.err
python tools/bench_summary.py $O/r04l_*.json

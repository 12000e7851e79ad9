# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "halo_3x3" 2>&1 | tail -3
for i in 1 2; do
  python tools/conv_bench.py --layer inc_4a_3x3 --which dgrad --flags 12 --iters 300
  DIN_HALO_BN80=0 python tools/conv_bench.py --layer inc_4a_3x3 --which dgrad --flags 12 --iters 300
done
timeout 600 python bench.py $B > $O/r04g_b32_bn80.json 2> $O/r04g_b32_bn80.err
DIN_HALO_BN80=0 timeout 600 python bench.py $B > $O/r04g_b32_bn96.json 2> $O/r04g_b32_bn96.err
timeout 600 python bench.py $B > $O/r04g_b32_bn80b.json 2> $O/r04g_b32_bn80b.err
DIN_HALO_BN80=0 timeout 600 python bench.py $B > $O/r04g_b32_bn96b.json 2> $O/r04g_b32_bn96b.err
python tools/bench_summary.py $O/r04g_*.json

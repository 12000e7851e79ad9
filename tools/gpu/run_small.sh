# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "t96_ or dgrad_x or mixed6a" 2>&1 | tail -5
timeout 600 python bench.py $B --global-batch 4 > $O/r04d_b4_t96.json 2> $O/r04d_b4_t96.err
DIN_CONV_TILE96=0 timeout 600 python bench.py $B --global-batch 4 > $O/r04d_b4_t128.json 2> $O/r04d_b4_t128.err
timeout 600 python bench.py $B --global-batch 4 > $O/r04d_b4_t96b.json 2> $O/r04d_b4_t96b.err
DIN_CONV_TILE96=0 timeout 600 python bench.py $B --global-batch 4 > $O/r04d_b4_t128b.json 2> $O/r04d_b4_t128b.err
DIN_WGRAD_STREAM=1 timeout 600 python bench.py $B --global-batch 4 > $O/r04d_b4_side.json 2> $O/r04d_b4_side.err
timeout 600 python bench.py $B --global-batch 4 --graph on > $O/r04d_b4_graph.json 2> $O/r04d_b4_graph.err
timeout 600 python bench.py $B --global-batch 8 > $O/r04d_b8_t96.json 2> $O/r04d_b8_t96.err
timeout 600 python bench.py $B --global-batch 16 > $O/r04d_b16.json 2> $O/r04d_b16.err
timeout 600 python bench.py $B > $O/r04d_b32.json 2> $O/r04d_b32.err
python tools/bench_summary.py $O/r04d_*.json

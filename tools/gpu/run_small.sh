# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "unsearched or lowp_linear" -s 2>&1 | grep -E "un-searched|passed|failed|Error" | cut -c1-400
timeout 600 python bench.py $B --global-batch 4 > $O/r04f_b4_direct.json 2> $O/r04f_b4_direct.err
DIN_WGRAD_DIRECT=0 timeout 600 python bench.py $B --global-batch 4 > $O/r04f_b4_reduce.json 2> $O/r04f_b4_reduce.err
timeout 600 python bench.py $B --global-batch 4 > $O/r04f_b4_direct2.json 2> $O/r04f_b4_direct2.err
DIN_WGRAD_DIRECT=0 timeout 600 python bench.py $B --global-batch 4 > $O/r04f_b4_reduce2.json 2> $O/r04f_b4_reduce2.err
python tools/bench_summary.py $O/r04f_*.json

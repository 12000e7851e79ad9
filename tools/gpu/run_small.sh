# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
B="--steps 20 --no-extras --no-cpu-baseline"
DIN_SINGLE_DEVICE=1 DIN_DIST_BACKEND=gloo DIN_CHECK_ALLREDUCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 2 --global-batch 8 --no-cpu-baseline 2>&1 | grep -E "allreduce check|metric|Error|error" | cut -c1-200
for i in 1 2; do
  timeout 600 python bench.py $B --global-batch 4 > $O/r04p_b4_plain$i.json 2> $O/r04p_b4_plain$i.err
  timeout 600 python bench.py $B --global-batch 4 --force-buckets 2>$O/r04p_b4_buckets$i.err | grep '"metric"' > $O/r04p_b4_buckets$i.json
done
tail -3 $O/r04p_b4_buckets1.err
python tools/bench_summary.py $O/r04p_*.json
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "train_net or captured or checkpoint or adam or trainer" 2>&1 | tail -3

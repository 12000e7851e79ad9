# scratch script for one-off gpurun calls during tuning (see lease.sh for the repeatable steps)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
for c in 4 32; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --global-batch $c --steps 20 --warmup 5 \
      --no-cpu-baseline --force-buckets 2>&1 | grep '"metric"' > $O/r04r_bench_inv3_bf16_b${c}_forced_buckets.json
done
python tools/bench_summary.py $O/r04r_*.json
DIN_SINGLE_DEVICE=1 DIN_DIST_BACKEND=gloo DIN_CHECK_ALLREDUCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 2 --global-batch 8 --no-cpu-baseline > $O/r04r_two_ranks_one_device.log 2>&1
tail -3 $O/r04r_two_ranks_one_device.log | cut -c1-200

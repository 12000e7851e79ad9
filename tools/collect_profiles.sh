#!/bin/bash
# Round-N measurement pass on one MI355X: every bench workload, the rocprofv3 kernel-trace summaries of the default and the 4-clip run, the
# two PMC traffic passes, the per-layer table and the two-ranks-on-one-device dry run of the multi-GPU path.  Writes gpurun_out/<tag>/;
# copy what should be judged into profiles/.
#   usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 python bench.py "$@" > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log > $OUT/bench_$name.json; cut -c1-160 $OUT/bench_$name.json; }
run inv3_bf16_b32 --per-layer $OUT/inv3_bf16_per_layer.txt
run inv3_bf16_b4 --global-batch 4 --no-cpu-baseline --no-extras --per-layer $OUT/inv3_bf16_b4_per_layer.txt
run inv3_bf16_b32_host_images --host-images --no-cpu-baseline
run inv3_bf16_b32_bn_batch --bn-mode batch --no-cpu-baseline
run inv3_bf16_b32_forward_only --forward-only --no-cpu-baseline
run inv3_bf16_b32_lite128 --lite-dim 128 --no-cpu-baseline
run inv3_bf16_b32_hier_t10 --hierarchical --no-cpu-baseline
run collective_vgg16_bf16_b8 --workload collective_bf16 --no-cpu-baseline
run collective_vgg16_fp32_b4 --workload collective_fp32 --global-batch 4 --no-cpu-baseline
run inv3_fp32_b8 --workload inv3_fp32 --global-batch 8 --no-cpu-baseline
run vgg16_bf16_b32 --workload vgg16_bf16 --no-cpu-baseline
run vgg16_fp32_b8 --workload vgg16_fp32 --global-batch 8 --no-cpu-baseline
run tce_vgg16_bf16_t10_b8 --workload vgg16_bf16 --tce --global-batch 8
run vgg16_bf16_t10_b8 --workload vgg16_bf16 --frames 10 --global-batch 8 --no-cpu-baseline
# gradient-bucket path in one process (world size 1, RCCL): hook + flat buckets + async all_reduce + views, no wire
for c in 4 32; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --global-batch $c --steps 20 --warmup 5 \
      --no-cpu-baseline --force-buckets 2>&1 | grep '"metric"' > $OUT/bench_inv3_bf16_b${c}_forced_buckets.json
  timeout 600 python bench.py --global-batch $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_inv3_bf16_b${c}_20steps.json
  cut -c75-170 $OUT/bench_inv3_bf16_b${c}_20steps.json $OUT/bench_inv3_bf16_b${c}_forced_buckets.json
done
# rocprofv3 kernel-trace summaries (same command as the default bench line, 5 timed + 2 warm-up steps)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_b32 -o x -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras > $OLDPWD/$OUT/prof_b32.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_b4 -o x -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --global-batch 4 > $OLDPWD/$OUT/prof_b4.log 2>&1)
find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/inv3_bf16_b32_kernel_stats.csv \;
find $OUT/prof_b4 -name "*kernel_stats.csv" -exec cp {} $OUT/inv3_bf16_b4_kernel_stats.csv \;
# HBM traffic per kernel: two separate PMC passes (kernel-trace only)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o x -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o x -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/pmc_write.log 2>&1)
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" $OUT/pmc_traffic.json
rm -rf $OUT/prof_b32 $OUT/prof_b4 $OUT/pmc_fetch $OUT/pmc_write
# multi-GPU path, dry: two ranks share this one device (gloo), every rank checks that all ranks hold identical averaged gradients
# (round 5: launched PLAINLY, the way the driver calls it -- bench.py starts its own two ranks)
DIN_SINGLE_DEVICE=1 DIN_DIST_BACKEND=gloo DIN_CHECK_ALLREDUCE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --global-batch 8 --no-cpu-baseline > $OUT/two_ranks_one_device.log 2>&1
tail -4 $OUT/two_ranks_one_device.log | cut -c1-300
ls -la $OUT | head -50

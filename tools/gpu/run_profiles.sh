cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/p
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/p/bench_inv3_bf16.json 2> gpurun_out/p/bench_inv3_bf16.err
timeout 900 python bench.py --steps 3 --warmup 1 --workload vgg16_bf16 --no-cpu-baseline > gpurun_out/p/bench_vgg16_bf16.json 2> gpurun_out/p/bench_vgg16_bf16.err
timeout 900 python bench.py --steps 3 --warmup 1 --workload vgg16_fp32 --global-batch 8 --no-cpu-baseline > gpurun_out/p/bench_vgg16_fp32.json 2> gpurun_out/p/bench_vgg16_fp32.err
timeout 900 python bench.py --steps 3 --warmup 1 --workload inv3_fp32 --global-batch 8 --no-cpu-baseline > gpurun_out/p/bench_inv3_fp32.json 2> gpurun_out/p/bench_inv3_fp32.err
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/p/prof_bench.log 2>&1
tail -c 600 gpurun_out/p/*.json
timeout 600 python bench.py --steps 10 --warmup 2 --global-batch 4 --no-cpu-baseline > gpurun_out/p/bench_inv3_bf16_b4.json 2> gpurun_out/p/bench_inv3_bf16_b4.err
tail -c 300 gpurun_out/p/bench_inv3_bf16_b4.json

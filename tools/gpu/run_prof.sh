cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --per-layer gpurun_out/per_layer.txt 2>&1 | tail -1 | cut -c1-200
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
tail -1 gpurun_out/prof_bench.log | cut -c1-200

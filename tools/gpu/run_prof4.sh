cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o r01 -- python bench.py --steps 5 --warmup 2 --global-batch 4 --no-cpu-baseline > gpurun_out/prof4_bench.log 2>&1
tail -1 gpurun_out/prof4_bench.log | cut -c1-200

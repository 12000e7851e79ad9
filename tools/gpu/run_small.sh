timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_last.json
cut -c1-700 gpurun_out/bench_last.json
rm -rf gpurun_out/prof
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
rm -f gpurun_out/prof/*kernel_trace.csv

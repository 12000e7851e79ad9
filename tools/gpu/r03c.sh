O=gpurun_out/r03c; mkdir -p $O; export TMPDIR=/tmp
(time python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_inv3_bf16_b32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_b32 -o x -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras > $OLDPWD/$O/prof_b32.log 2>&1)
find $O/prof_b32 -name "*kernel_stats.csv" -exec cp {} $O/inv3_bf16_b32_kernel_stats.csv \;
rm -rf $O/prof_b32
STRESS_SECONDS=900 tools/stress_gpu_suite.sh run 3 > $O/stress_loop.log 2>&1
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-200 $O/bench_inv3_bf16_b32.json; head -5 $O/inv3_bf16_b32_kernel_stats.csv | cut -c1-200; cat $O/stress_loop.log

set -u
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_b32 -o x -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras > $OLDPWD/$OUT/prof_b32.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_b4 -o x -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --global-batch 4 > $OLDPWD/$OUT/prof_b4.log 2>&1)
find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/inv3_bf16_b32_kernel_stats.csv \;
find $OUT/prof_b4 -name "*kernel_stats.csv" -exec cp {} $OUT/inv3_bf16_b4_kernel_stats.csv \;
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o x -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o x -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/pmc_write.log 2>&1)
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" $OUT/pmc_traffic.json
rm -rf $OUT/prof_b32 $OUT/prof_b4 $OUT/pmc_fetch $OUT/pmc_write
python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
ls -la $OUT

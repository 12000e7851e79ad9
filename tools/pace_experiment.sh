for pace in 0 1; do for L in inc_4a_3x3 inc_6a_3x3 inc_6e_7x1 inc_6b_1x1; do DIN_WGRAD_PACE=$pace timeout 100 python tools/conv_bench.py --layer $L --which wgrad 2>&1 | tail -1 | sed "s/^/pace=$pace /"; done; done
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pipe or wgrad" 2>&1 | tail -2
for pace in 0 1; do DIN_WGRAD_PACE=$pace timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180; done
export TMPDIR=/tmp; cd /tmp; for pace in 0 1; do DIN_WGRAD_PACE=$pace timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf$pace -o x -- python /root/repo/tools/conv_bench.py --layer inc_4a_3x3 --which wgrad --iters 3 > /dev/null 2>&1; python - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pf$pace/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "wgrad_pipe" in r["Kernel_Name"]]
print("pace=$pace FETCH x2 GB per launch", 2e3*sum(v)/len(v)/1e9)
PY
done

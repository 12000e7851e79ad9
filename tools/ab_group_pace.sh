#!/bin/bash
# A/B of sibling pacing inside the grouped weight-gradient launch: step time (alternating runs) and HBM fetch bytes of the group kernel.
OUT=${1:-gpurun_out/grouppace}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for i in 1 2; do
  for P in 1 0; do
    DIN_OPTIONS_FROM_ENV=1 DIN_WGRAD_PACE=$P python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/b32_pace${P}_$i.log 2>&1
    DIN_OPTIONS_FROM_ENV=1 DIN_WGRAD_PACE=$P python bench.py --global-batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/b4_pace${P}_$i.log 2>&1
  done
done
for P in 1 0; do
  (cd /tmp && DIN_OPTIONS_FROM_ENV=1 DIN_WGRAD_PACE=$P timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch_$P -o x -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/$OUT/pmc_fetch_$P.log 2>&1)
  F=$(find $OUT/pmc_fetch_$P -name "*counter_collection.csv" | head -1)
  python - "$F" $P <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "conv_wgrad_pipe" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("conv_wgrad_pipe")[1].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"pace={sys.argv[2]} conv_wgrad_pipe{k}: {len(v)} launches, fetch {2e3 * sum(v) / len(v) / 1e6:8.1f} MB per launch (FETCH_SIZE x 2, gfx950 wide-read correction)")
PY
  rm -rf $OUT/pmc_fetch_$P
done
for f in $OUT/b*.log; do echo -n "$f "; grep '"metric"' $f | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done

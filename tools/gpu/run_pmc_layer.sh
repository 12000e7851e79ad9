# SQ / TCC counters of ONE conv launch shape (tools/conv_bench.py), one rocprofv3 --pmc pass per counter group, kernel-trace only.
#   bash tools/gpu/run_pmc_layer.sh <layer> <fwd|dgrad|wgrad> <kernel-name substring>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=${1:-inc_6e_7x1}
W=${2:-fwd}
K=${3:-conv_gather}
OUT=gpurun_out/pmc_${L}_${W}
rm -rf $OUT; mkdir -p $OUT
python tools/conv_bench.py --layer $L --which $W --iters 20 2>&1 | grep -v amdgpu.ids | tee $OUT/timing.txt
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/g$i -o p -- python tools/conv_bench.py --layer $L --which $W --iters 3 > $OUT/log_g$i.txt 2>&1; echo "rc=$?"
done
K=$K OUT=$OUT python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
for f in sorted(glob.glob(os.environ['OUT'] + '/g*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if os.environ['K'] in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(f.split('/')[-2], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc5
L=${1:-inc_6e_7x1}
for w in fwd; do
python tools/conv_bench.py --layer $L --which $w --iters 20 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer inc_6e_1x7 --which $w --iters 20 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer inc_6e_1x1_768 --which $w --iters 20 2>&1 | grep -v amdgpu.ids
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc5/$w$i -o p -- python tools/conv_bench.py --layer $L --which $w --iters 3 > gpurun_out/pmc5/log_$w$i.txt 2>&1; echo "rc=$?"
done
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc5/*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'conv_gather' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(f.split('/')[-2], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY

#!/bin/bash
# SQ / TCC counters of one conv kernel on one layer shape: separate rocprofv3 --pmc passes (kernel-trace only, as the microarch guide
# prescribes) over tools/conv_bench.py, summarised per launch by tools/pmc_layer.py.
#   usage: tools/pmc_layer.sh <layer> <fwd|dgrad|wgrad> <kernel-name-substring> <out-dir> [ENV=VAL ...]
set -u
LAYER=$1; WHICH=$2; KSUB=$3; OUT=$4; shift 4
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for GROUP in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OLDPWD/$OUT/p$i" -o x -- \
      python "$OLDPWD/tools/conv_bench.py" --layer "$LAYER" --which "$WHICH" --iters 3 > "$OLDPWD/$OUT/p$i.log" 2>&1)
done
python tools/pmc_layer.py "$OUT" "$KSUB" | tee "$OUT/summary.txt"

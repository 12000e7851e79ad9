#!/bin/bash
# First command of a multi-GPU lease: proves that RCCL really ran with N ranks (one process per GPU over xGMI) before any number is
# quoted.  RCCL has never run with more than one rank on this code (single-GPU leases only), so the data-parallel path is
# correct by construction + gloo world-2 CPU tests; this script is the missing hardware check.
#   usage: tools/rccl_selfcheck.sh [N=8] [global_batch=32]
# It launches bench.py as the driver does, with
#   DIN_CHECK_ALLREDUCE=1   every step: all ranks hold the same averaged-gradient checksum (MIN == MAX over ranks), and the line
#                           "rccl selfcheck: backend nccl, world_size N, devices [...]" is printed after asserting world_size == N,
#                           backend == nccl and N distinct devices;
# and greps for both.  Exit status 0 only if every check passed and the JSON line says n_gpus == N.
set -u
cd "$(dirname "$0")/.."
N=${1:-8}
GB=${2:-32}
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$(mktemp)
DIN_CHECK_ALLREDUCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus "$N" --steps 3 --warmup 2 --global-batch "$GB" --no-extras --no-cpu-baseline > "$LOG" 2>&1
rc=$?
grep -E "rccl selfcheck|allreduce check ok" "$LOG" | head -8
tail -1 "$LOG" | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
assert d['n_gpus'] == $N, d['n_gpus']
print('bench line: n_gpus', d['n_gpus'], 'value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], '|', d['config']['includes'])
" || rc=1
grep -q "rccl selfcheck: backend nccl, world_size $N," "$LOG" || { echo "MISSING: rccl selfcheck line for world_size $N"; rc=1; }
grep -q "allreduce check ok" "$LOG" || { echo "MISSING: allreduce checksum line"; rc=1; }
[ $rc -ne 0 ] && tail -30 "$LOG"
rm -f "$LOG"
echo "rccl_selfcheck: $([ $rc -eq 0 ] && echo PASS || echo FAIL)"
exit $rc

#!/bin/bash
# din_walk_bwd_kernel with one wave per position (the round-1 launch: DIN_WALK_BWD_GROUPS=9 at T x N = 36) and with the default grouping
cd /tmp && export TMPDIR=/tmp
for g in 9 0; do
  DIN_WALK_BWD_GROUPS=$g rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wg_$g -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== DIN_WALK_BWD_GROUPS=$g"
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/wg_$g -name "*kernel_stats.csv" | head -1)
  grep -E "din_walk" $f | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:80], r[1], r[3])"
done

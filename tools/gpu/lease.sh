#!/bin/bash
# One parameterised GPU-lease script (replaces the 18 one-shot tools/gpu/r03*.sh of round 3).
#   usage (on the GPU box, via gpurun):  tools/gpu/lease.sh <tag> <step> [<step> ...]
# steps:  suite      full `pytest -m gpu` (no -x: every failure is listed) -> gpurun_out/<tag>_suite.log, margins -> <tag>_test_margins.txt
#         suitex     the driver's form (`-x -q`)
#         repeat:N   the full suite N times (-x), one summary line each -> <tag>_suite_repeat.log
#         bench      default bench line -> <tag>_bench.json ;  bench:<args...> with ',' for spaces, e.g. bench:--bn-mode,batch
#         stats      rocprofv3 --kernel-trace --stats of the default bench -> <tag>_kernel_stats.csv
#         smoke      __graft_entry__.smoke()
#         seq:<args> launch sequence of one step (tools/launch_sequence.py over a rocprofv3 kernel trace) -> <tag>_launch_sequence_<args>.txt
#         env:K=V / unset:K   environment for the following steps;   sh:<command with , for spaces>   anything else (last lines shown)
set -u
cd "$(dirname "$0")/../.."
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
export DIN_OPTIONS_FROM_ENV=1        # env:K=V steps below reach the library as options (ABI 8: it never reads the environment itself)
for step in "$@"; do
  case "$step" in
    suite)   timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/${TAG}_suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/${TAG}_suite.log
             cp $OUT/test_margins.txt $OUT/${TAG}_test_margins.txt 2>/dev/null ;;
    suitex)  timeout 3000 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/${TAG}_suitex.log 2>&1; echo "suitex rc=$?"; tail -3 $OUT/${TAG}_suitex.log
             cp $OUT/test_margins.txt $OUT/${TAG}_test_margins.txt 2>/dev/null ;;
    repeat:*) n=${step#repeat:}; : > $OUT/${TAG}_suite_repeat.log
             for i in $(seq 1 $n); do
               r=$(timeout 3000 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1)
               echo "run $i on $(hostname): $r" | tee -a $OUT/${TAG}_suite_repeat.log
               cp $OUT/test_margins.txt $OUT/${TAG}_test_margins_run$i.txt 2>/dev/null
             done ;;
    bench)   timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json ;;
    bench:*) a=${step#bench:}; n=$(echo "$a${BTAG:-}" | tr -c 'a-zA-Z0-9' '_'); timeout 900 python bench.py ${a//,/ } > $OUT/${TAG}_bench_$n.json 2> $OUT/${TAG}_bench_$n.err
             echo "bench $a rc=$?"; cat $OUT/${TAG}_bench_$n.json ;;
    stats)   rm -rf /tmp/prof; (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-extras > /tmp/prof.log 2>&1)
             f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -60 "$f" > $OUT/${TAG}_kernel_stats.csv; tail -2 /tmp/prof.log ;;
    stats:*) a=${step#stats:}; n=$(echo "$a" | tr -c 'a-zA-Z0-9' '_'); rm -rf /tmp/prof
             (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-extras ${a//,/ } > /tmp/prof.log 2>&1)
             f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -60 "$f" > $OUT/${TAG}_kernel_stats_$n.csv; tail -2 /tmp/prof.log ;;
    seq:*)   a=${step#seq:}; n=$(echo "$a" | tr -c 'a-zA-Z0-9' '_'); rm -rf /tmp/tr
             (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline ${a//,/ } > /tmp/tr.log 2>&1)
             python tools/launch_sequence.py /tmp/tr > $OUT/${TAG}_launch_sequence_$n.txt 2>&1; head -4 $OUT/${TAG}_launch_sequence_$n.txt ;;
    env:*)   export "${step#env:}"; echo "export ${step#env:}" ;;
    unset:*) unset "${step#unset:}" ;;
    sh:*)    c=${step#sh:}; echo "+ ${c//,/ }"; bash -c "${c//,/ }" 2>&1 | tail -${TAILN:-5} ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
    *)       echo "unknown step $step" ;;
  esac
done

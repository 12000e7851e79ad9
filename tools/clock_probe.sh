#!/bin/bash
# Samples rocm-smi (sclk / mclk / socket power / temperature) every 0.3 s while a workload runs: is the chip clock- or power-limited under it?
# usage: tools/clock_probe.sh OUT -- command ...
OUT=$1; shift; shift
"$@" > $OUT.run.log 2>&1 &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
  rocm-smi -d 0 --showclocks --showpower --showtemp --showuse --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.3
done
wait $PID

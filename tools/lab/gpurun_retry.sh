#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 = "no box or slot free right now, nothing charged" — the ONLY
# case that is retried (every 2 min, up to 30 times). Any other outcome (the command ran) is final.
#   usage: tools/lab/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3

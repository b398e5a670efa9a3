#!/bin/bash
# gpurun with a retry on "no box / no slot free right now" (exit 3: nothing ran, nothing was charged).
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'      (never retries a command that RAN)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged).
#   scripts/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3

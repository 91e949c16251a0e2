#!/bin/bash
# local helper: submit one gpurun call, retrying only while the pod has no free GPU slot / box (exit 3: nothing ran, nothing charged)
# usage: scripts/gpu_retry.sh <timeout_s> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3

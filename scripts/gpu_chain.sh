#!/bin/bash
# Run a chain of GPU steps on the box: each under its own `timeout -k 10`; an ordinary failure (assertion, rc 1) lets the chain
# continue, a step that was killed / timed out (rc 124, 137 or any signal) ends the chain -- no further GPU step after a hang.
# usage: bash scripts/gpu_chain.sh "<secs>|<logname>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    secs=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
    echo "=== [$name] $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] rc=$rc"; tail -n 6 "gpurun_out/$name.log" | cut -c1-300
    if [ $rc -ge 124 ]; then echo "step killed (rc=$rc): chain stops"; exit $rc; fi
done
exit 0

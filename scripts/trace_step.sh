#!/bin/bash
# kernel trace of a short bench run -> gpurun_out/trace/ (per-launch durations)
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/trace.log 2>&1
echo rc=$?

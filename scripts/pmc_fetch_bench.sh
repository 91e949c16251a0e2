#!/bin/bash
# bench line + FETCH_SIZE pass of the bench (quick traffic check of a kernel change)
R=$PWD; export TMPDIR=/tmp
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>$R/gpurun_out/bench_err.log | tail -1 | cut -c1-260
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fb -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_fb.log 2>&1 || echo "fetch pass failed"
echo done

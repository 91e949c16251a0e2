#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the microbenchmark kernels (separate passes)
R=$PWD; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fmb/$c -- python3 $R/scripts/mb_res100.py > $R/gpurun_out/pmc_fmb_$c.log 2>&1 || echo "pass $c failed"
done
echo done

#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r05_sa_step; rm -rf $OUT; mkdir -p $OUT
python3 $R/scripts/ab_sa_fused.py 10 bf16 > $OUT/ab.txt 2>&1 || { echo ab failed; tail -5 $OUT/ab.txt; exit 1; }
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_sa.py bf16 > $OUT/trace.log 2>&1 || { echo trace failed; exit 1; }
python3 $R/scripts/sa_window.py $OUT/trace 8 14 > $OUT/window.txt
python3 $R/scripts/trace_table.py $OUT/trace 5 40 > $OUT/table.txt
find $OUT -name "*_kernel_trace.csv" -size +20M -delete

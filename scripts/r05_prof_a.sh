#!/bin/bash
# per-launch traces of the bf16 step, the cfg1 step, predict b1 (bf16) and the SA step
R=$PWD; OUT=$R/gpurun_out/r05_prof_a; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/bf16 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype bf16 > $OUT/bf16.log 2>&1 || { echo bf16 failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/bf16 4 20 > $OUT/bf16_table.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/cfg1 -- python3 $R/scripts/prof_cfg1.py cfg1 > $OUT/cfg1.log 2>&1 || { echo cfg1 failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/cfg1 20 5 > $OUT/cfg1_table.txt
UNET_DTYPE=bf16 rocprofv3 --kernel-trace --output-format csv -d $OUT/b1 -- python3 $R/scripts/prof_cfg1.py b1 > $OUT/b1.log 2>&1 || { echo b1 failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/b1 20 5 > $OUT/b1_table.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/sa -- python3 $R/scripts/prof_sa.py f32 > $OUT/sa.log 2>&1 || { echo sa failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/sa 5 50 > $OUT/sa_table.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/sabf -- python3 $R/scripts/prof_sa.py bf16 > $OUT/sabf.log 2>&1 || { echo sabf failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/sabf 5 30 > $OUT/sabf_table.txt
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
echo done

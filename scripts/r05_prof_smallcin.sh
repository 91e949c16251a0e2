R=$PWD; OUT=$R/gpurun_out/smallcin_prof; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
UNET_DTYPE=bf16 rocprofv3 --kernel-trace --output-format csv -d $OUT/b16 -- python3 $R/scripts/prof_cfg1.py b16 > $OUT/b16.log 2>&1 || { echo b16 failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/b16 20 5 > $OUT/b16_table.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/f32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/f32.log 2>&1 || { echo f32 failed; exit 1; }
python3 $R/scripts/trace_table.py $OUT/f32 4 20 > $OUT/f32_table.txt
find $OUT -name "*_kernel_trace.csv" -delete
grep -h "smallcin\|1, 1, 4, 1" $OUT/*_table.txt

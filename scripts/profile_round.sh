#!/bin/bash
# GPU-side evidence run for profiles/: bench line, rocprofv3 kernel stats, two separate PMC passes (HBM traffic), SQ pass.
# usage (from the repo root, through gpurun): bash scripts/profile_round.sh <tag> [f32|bf16]
TAG=${1:-r01_x}
DT=${2:-f32}
R=$PWD; OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
python -m unet_amd.build --hash > $OUT/code_hash.txt
python bench.py --dtype $DT > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.json | cut -c1-200
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --dtype $DT > $OUT/stats.log 2>&1 || { echo "stats pass failed"; exit 1; }
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $DT > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $DT > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $DT > $OUT/pmc_sq.log 2>&1 || { echo "sq pass failed"; exit 1; }
echo sq done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $DT > $OUT/pmc_lds.log 2>&1 || { echo "lds pass failed"; exit 1; }
echo lds done

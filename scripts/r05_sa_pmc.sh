#!/bin/bash
# counters of the fused SelfAttention launches alone (scripts/sa_kernels.py): traffic past the L2, MFMA busy, LDS
R=$PWD; OUT=$R/gpurun_out/r05_sa_pmc; rm -rf $OUT; mkdir -p $OUT
python3 $R/scripts/sa_kernels.py 10 > $OUT/timing.txt 2>&1 || { echo timing failed; cat $OUT/timing.txt; exit 1; }
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/scripts/sa_kernels.py 2 > $OUT/fetch.log 2>&1 || { echo fetch failed; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/scripts/sa_kernels.py 2 > $OUT/sq.log 2>&1 || { echo sq failed; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/lds -- python3 $R/scripts/sa_kernels.py 2 > $OUT/lds.log 2>&1 || { echo lds failed; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $R/scripts/sa_kernels.py 2 > $OUT/tcc.log 2>&1 || echo tcc failed
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "") or "/root/repo/gpurun_out/r05_sa_pmc"
for d in ("fetch", "sq", "lds", "tcc"):
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "sa_" not in k: continue
        k = k.split("(")[0].split("::")[-1][:28]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(d, k, {c: round(x / cnt[(k, c)], 1) for c, x in v.items()})
PY

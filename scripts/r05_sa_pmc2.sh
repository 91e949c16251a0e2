#!/bin/bash
# instruction-mix counters of the fused SelfAttention launches alone (scripts/sa_kernels.py): what the SIMDs issue next to the MFMAs
R=$PWD; OUT=$R/gpurun_out/r05_sa_pmc2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/mix1 -- python3 $R/scripts/sa_kernels.py 2 > $OUT/mix1.log 2>&1 || { echo mix1 failed; tail -3 $OUT/mix1.log; }
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/mix2 -- python3 $R/scripts/sa_kernels.py 2 > $OUT/mix2.log 2>&1 || { echo mix2 failed; tail -3 $OUT/mix2.log; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/act -- python3 $R/scripts/sa_kernels.py 2 > $OUT/act.log 2>&1 || { echo act failed; tail -3 $OUT/act.log; }
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/wait -- python3 $R/scripts/sa_kernels.py 2 > $OUT/wait.log 2>&1 || { echo wait failed; tail -3 $OUT/wait.log; }
python3 - <<'PY'
import csv, glob, collections, re, os
out = os.environ.get("OUT2", "/root/repo/gpurun_out/r05_sa_pmc2")
for d in ("mix1", "mix2", "act", "wait"):
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"(sa_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "fwd" in k or "bwd" in k: print(d, k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY

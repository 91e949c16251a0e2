#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q --durations=12 > $O/r04_d_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -22 $O/r04_d_tests.log
[ $rc -eq 0 ] || exit 1
for dt in f32 bf16; do for on in 0 1; do UNET_CONV1X1_GEMM=$on python scripts/ab_conv1x1.py $dt; done; done > $O/r04_d_conv1x1.log 2>&1; grep gemm1x1 $O/r04_d_conv1x1.log
python bench.py > $O/r04_d_bench.json 2> $O/r04_d_bench.err || { echo bench failed; tail -5 $O/r04_d_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_d_bench.json").read().strip().splitlines()[-1]); s=d.get("secondary",{})
print("f32", d["value"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_launch_ms"])
for k,v in s.items():
    if isinstance(v,dict) and "value" in v: print(k, v["value"], v.get("roofline",{}).get("mfma_frac_of_bf16_peak"))
    elif isinstance(v,dict): print(k, {kk:(vv.get("value") if isinstance(vv,dict) else vv) for kk,vv in v.items()})
PY

#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_g_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r04_g_tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/r04_g_tests.log | tail -20; exit 1; }
python bench.py > $O/r04_g_bench.json 2> $O/r04_g_bench.err || { echo bench failed; tail -5 $O/r04_g_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_g_bench.json").read().strip().splitlines()[-1]); s=d.get("secondary",{})
print("f32", d["value"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_launch_ms"], "mem", d["hbm_bytes_allocated"])
for k,v in s.items():
    if isinstance(v,dict) and "value" in v: print(k, v["value"], {kk:vv for kk,vv in v.get("roofline",{}).items() if kk in ("mfma_frac_of_bf16_peak","probed_value","frac","avg_launch_ms")})
    elif isinstance(v,dict): print(k, {kk:(vv.get("value") if isinstance(vv,dict) else vv) for kk,vv in v.items()})
print(d.get("cpu_baseline"))
PY

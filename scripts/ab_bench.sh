#!/bin/bash
# A/B of one environment knob inside ONE gpurun call (same box): scripts/ab_bench.sh KNOB tag   -> gpurun_out/<tag>_{on,off}.json
KNOB=$1; TAG=$2; shift 2
for v in 1 0 1 0; do
  env $KNOB=$v python bench.py --no-cpu-baseline "$@" > gpurun_out/${TAG}_$v.json.tmp 2>> gpurun_out/${TAG}.err || exit 1
  cat gpurun_out/${TAG}_$v.json.tmp >> gpurun_out/${TAG}_$v.jsonl
done
python - <<PY
import json
for v in (1, 0):
    for line in open("gpurun_out/${TAG}_%d.jsonl" % v):
        d = json.loads(line)
        s = d.get("secondary", {})
        print("${KNOB}=%d" % v, "f32", d["value"], "| bf16", s.get("bf16", {}).get("value"), "| predict b16", s.get("predict_b16", {}).get("f32", {}).get("value"), s.get("predict_b16", {}).get("bf16", {}).get("value"),
              "| b1", s.get("predict_b1", {}).get("f32", {}).get("value"), s.get("predict_b1", {}).get("bf16", {}).get("value"), "| cfg1", s.get("cfg1", {}).get("value"),
              "| cfg5", s.get("cfg5", {}).get("f32", {}).get("value"), s.get("cfg5", {}).get("bf16", {}).get("value"))
PY

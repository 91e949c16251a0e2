#!/bin/bash
# round-4 first GPU call: GPU tests, default bench (side-stream weight gradients on), the same with UNET_WGRAD_STREAM=0, 6-rank gloo rehearsal
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_a_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_a_tests.log
python bench.py > $O/r04_a_bench.json 2> $O/r04_a_bench.err || { echo bench failed; tail -5 $O/r04_a_bench.err; exit 1; }
UNET_WGRAD_STREAM=0 python bench.py --no-cpu-baseline > $O/r04_a_bench_noside.json 2> $O/r04_a_bench_noside.err || { echo bench noside failed; exit 1; }
python - <<'PY'
import json
for f in ("r04_a_bench.json","r04_a_bench_noside.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); s=d.get("secondary",{})
    g=lambda k,*p: (lambda v: v if not isinstance(v,dict) else v.get("value", v.get("error")))(eval("s"+"".join(f"[{q!r}]" for q in (k,)+p)) if k in s else None)
    print(f, "f32", d["value"], "frac", d["roofline"]["frac"], "bf16", g("bf16"), "cfg1", g("cfg1"), "cfg4", g("cfg4"), "sa", s.get("sa_on"), "p16", s.get("predict_b16",{}).get("f32",{}).get("value"), "cfg5", s.get("cfg5",{}).get("f32",{}).get("value") if isinstance(s.get("cfg5"),dict) else None)
PY
UNET_DIST_BACKEND=gloo UNET_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 6 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline > $O/r04_a_rehearsal6.json 2> $O/r04_a_rehearsal6.err; echo "rehearsal rc=$?"; tail -c 1200 $O/r04_a_rehearsal6.json

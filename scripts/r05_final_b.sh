#!/bin/bash
# end-of-round evidence, part B: the default bench line on the final code (pmc_traffic.json of part A committed) and the N > 1 rehearsal
O=gpurun_out
python bench.py > $O/r05_zz_bench.json 2> $O/r05_zz_bench.err || { echo bench failed; tail -5 $O/r05_zz_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_zz_bench.json").read().strip().splitlines()[-1]); s=d.get("secondary",{})
r=d["roofline"]
print("f32", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "avg_ms", r["avg_launch_ms"], "traffic", r["traffic"], r["traffic_source"], r["traffic_code_current"], "mem", d["hbm_bytes_allocated"])
for k,v in s.items():
    if isinstance(v,dict) and "value" in v: print(k, v["value"], {kk:vv for kk,vv in v.get("roofline",{}).items() if kk in ("mfma_frac_of_bf16_peak","probed_value","frac","avg_launch_ms","traffic","traffic_code_current")})
    elif isinstance(v,dict): print(k, {kk:(vv.get("value") if isinstance(vv,dict) else vv) for kk,vv in v.items()})
print(d.get("cpu_baseline"))
PY
UNET_DIST_BACKEND=gloo UNET_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 6 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline > $O/r05_zz_rehearsal6.json 2> $O/r05_zz_rehearsal6.err; echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_zz_rehearsal6.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','rccl_ranks','dist_backend')}, [x['allreduce_wait_ms_per_step'] for x in d['devices']])
c=d['secondary']['cfg5']; print(c['f32']['mask_checksum'], c['f32']['value'], c['bf16']['mask_checksum'], c['bf16']['value'])
PY

#!/bin/bash
# the N > 1 rehearsal alone: six gloo ranks on the one GPU of the box
O=gpurun_out
UNET_DIST_BACKEND=gloo UNET_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 6 --batch 2 --steps 3 --warmup 2 --no-cpu-baseline > $O/r05_zz_rehearsal6.json 2> $O/r05_zz_rehearsal6.err; echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_zz_rehearsal6.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','rccl_ranks','dist_backend')}, [x['allreduce_wait_ms_per_step'] for x in d['devices']])
c=d['secondary']['cfg5']; print(c['f32']['mask_checksum'], c['f32']['value'], c['bf16']['mask_checksum'], c['bf16']['value'])
PY

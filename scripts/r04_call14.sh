#!/bin/bash
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time, torch
sys.path.insert(0, '.')
import bench as B
dev = torch.device("cuda", 0)
for rep in range(2):
    for dt, b in (("f32", 16), ("bf16", 16), ("f32", 1), ("bf16", 1), ("bf16", 1), ("f32", 1)):
        r = B.predict_bench(dt, b, dev, iters=20 if b == 1 else 6)
        print(dt, b, r["value"], r["ms_per_batch"], flush=True)
PY

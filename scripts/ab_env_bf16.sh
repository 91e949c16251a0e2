#!/bin/bash
# A/B of environment settings on the bf16 step inside one gpurun call: scripts/ab_env_bf16.sh "VAR=a" "VAR=b" ...
for rep in 1 2; do
for kv in "$@"; do
  env $kv python bench.py --dtype bf16 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['ms_per_step'])"
done
done

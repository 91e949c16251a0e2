#!/bin/bash
# several values of one environment knob inside ONE gpurun call: scripts/ab_values.sh KNOB "v1 v2 ..." [bench args]
KNOB=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $KNOB=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('secondary',{})
print('$KNOB=$v', d['dtype'], d['value'], '| bf16', s.get('bf16',{}).get('value'), '| b16', s.get('predict_b16',{}).get('f32',{}).get('value'), s.get('predict_b16',{}).get('bf16',{}).get('value'), '| b1', s.get('predict_b1',{}).get('f32',{}).get('value'), '| cfg1', s.get('cfg1',{}).get('value'))"
done; done

#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "bn_fused" > $O/r04_l_t1.log 2>&1; rc=$?; echo "bn op tests rc=$rc"; tail -3 $O/r04_l_t1.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" $O/r04_l_t1.log | tail; exit 1; }
python -m pytest tests -m gpu -x -q > $O/r04_l_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r04_l_tests.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/r04_l_tests.log | tail -20; exit 1; }
for f in 0 1; do
  echo "UNET_FUSE_SMALL_BN=$f"
  UNET_FUSE_SMALL_BN=$f python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids | grep "graph=False"
  UNET_FUSE_SMALL_BN=$f UNET_DTYPE=bf16 python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids | grep "graph=False"
  UNET_FUSE_SMALL_BN=$f python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 step', d['value'])"
  UNET_FUSE_SMALL_BN=$f python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 step (probed)', d['value'])"
done | tee $O/r04_l_bn_ab.log

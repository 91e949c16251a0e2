#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q -k "conv1x1 or shuffle" > $O/r04_e_tests.log 2>&1; rc=$?; echo "op tests rc=$rc"; tail -5 $O/r04_e_tests.log
[ $rc -eq 0 ] || exit 1
python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py tests/test_bf16_gpu.py -x -q > $O/r04_e_tests2.log 2>&1; rc=$?; echo "model tests rc=$rc"; tail -5 $O/r04_e_tests2.log
[ $rc -eq 0 ] || exit 1
python scripts/ab_shuffle.py 2>&1 | grep -v amdgpu.ids | tee $O/r04_e_shuffle.log

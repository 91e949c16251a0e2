#!/bin/bash
set -o pipefail
O=gpurun_out
python scripts/ab_side2.py > $O/r04_c_side2.log 2>&1; cat $O/r04_c_side2.log | grep -v amdgpu.ids
python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -x -q -k "mixed or cfg1 or default_init or randomised or shipped" --durations=8 > $O/r04_c_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/r04_c_tests.log

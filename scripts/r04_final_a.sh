#!/bin/bash
# end-of-round evidence, part A: GPU tests, then the rocprofv3 passes of both storage types on the final code
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_k_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r04_k_tests.log
[ $rc -eq 0 ] || { grep -n "Error\|FAILED" $O/r04_k_tests.log | tail; exit 1; }
bash scripts/profile_round.sh r04_k f32 && bash scripts/profile_round.sh r04_k_bf16 bf16

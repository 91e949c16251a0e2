#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "t256" > $O/r04_b_tests.log 2>&1; echo "t256 op tests rc=$?"; tail -3 $O/r04_b_tests.log
UNET_T256_SLIVER=0 python scripts/ab_sliver.py > $O/r04_b_sliver.log 2>&1 && UNET_T256_SLIVER=1 python scripts/ab_sliver.py >> $O/r04_b_sliver.log 2>&1; cat $O/r04_b_sliver.log
python scripts/ab_side.py > $O/r04_b_side.log 2>&1; cat $O/r04_b_side.log

#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/ab_shuffle.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_p_shuffle_staged.log

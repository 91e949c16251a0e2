#!/bin/bash
python -m pytest tests/test_configs_gpu.py -x -q -k "cfg2_training_step" -s 2>&1 | grep -v amdgpu.ids | tail -14

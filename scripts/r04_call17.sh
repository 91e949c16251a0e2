#!/bin/bash
python -m pytest tests/test_raster_gpu.py -x -q -k "batch_invariant" 2>&1 | grep -v amdgpu.ids | tail -8

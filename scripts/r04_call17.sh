#!/bin/bash
python -m pytest tests -m gpu -x -q --durations=10 2>&1 | grep -v amdgpu.ids | tail -18

#!/bin/bash
python -m pytest tests/test_fullsize_gpu.py tests/test_raster_gpu.py -x -q 2>&1 | tail -4

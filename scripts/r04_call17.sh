#!/bin/bash
python -m pytest tests/test_model_gpu.py tests/test_learner_gpu.py -x -q 2>&1 | tail -3
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')
" 2>&1 | tail -2

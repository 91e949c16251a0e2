#!/bin/bash
O=gpurun_out
python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids | tee $O/r04_f_cfg1_graph.log
UNET_DTYPE=bf16 python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_f_cfg1_graph.log
bash scripts/r04_call7.sh 2>&1 | tee $O/r04_f_pmc1x1.log

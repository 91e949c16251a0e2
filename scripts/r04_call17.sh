#!/bin/bash
python scripts/layer_table.py 321287 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_q_layer_table_f32.txt; head -22 gpurun_out/r04_q_layer_table_f32.txt
UNET_DTYPE=bf16 python scripts/layer_table.py 321287 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_q_layer_table_bf16.txt; head -20 gpurun_out/r04_q_layer_table_bf16.txt

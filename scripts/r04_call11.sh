#!/bin/bash
python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q 2>&1 | tail -3
python scripts/host_profile.py f32 2>&1 | grep -v amdgpu.ids | head -40 | tee gpurun_out/r04_i_host_f32.log
python scripts/host_profile.py bf16 2>&1 | grep -v amdgpu.ids | head -3 | tee gpurun_out/r04_i_host_bf16.log
python scripts/bench_extra.py cfg1 predict 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_i_extra.log
UNET_DTYPE=bf16 python scripts/bench_extra.py cfg1 predict 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_i_extra.log

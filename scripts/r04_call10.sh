#!/bin/bash
python scripts/host_profile.py f32 2>&1 | grep -v amdgpu.ids | head -60 | tee gpurun_out/r04_h_host_f32.log
python scripts/host_profile.py bf16 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/r04_h_host_bf16.log

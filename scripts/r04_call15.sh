#!/bin/bash
for dt in f32 bf16; do UNET_CONV1X1_GEMM=1 python scripts/ab_conv1x1.py $dt; done 2>&1 | grep gemm1x1 | tee gpurun_out/r04_n_conv1x1_3sets.log
python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -x -q -k "conv1x1" 2>&1 | tail -2

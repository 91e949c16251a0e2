#!/bin/bash
for s in 0 1; do
  echo "UNET_WGRAD_STREAM=$s"
  UNET_WGRAD_STREAM=$s python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids
  UNET_WGRAD_STREAM=$s UNET_DTYPE=bf16 python scripts/bench_extra.py cfg1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_j_graph_side.log

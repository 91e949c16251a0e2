#!/bin/bash
# A/B of two builds of the library on ONE box: scripts/ab_build.sh "<hipcc flags A>" "<hipcc flags B>" <command...>
# builds with flags A, runs the command, builds with flags B, runs it, twice; output in gpurun_out/ab_build.log
FA=$1; FB=$2; shift 2
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then F="$FA"; else F="$FB"; fi
    UNET_EXTRA_HIPCC_FLAGS="$F" python -m unet_amd.build --force > gpurun_out/ab_build_make.log 2>&1 || exit 1
    echo "==== build $v ($F) rep $rep" >> gpurun_out/ab_build.log
    "$@" >> gpurun_out/ab_build.log 2>&1 || exit 1
  done
done
UNET_EXTRA_HIPCC_FLAGS="" python -m unet_amd.build --force > gpurun_out/ab_build_make.log 2>&1
cat gpurun_out/ab_build.log

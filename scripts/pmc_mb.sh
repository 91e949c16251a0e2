#!/bin/bash
# PMC passes over the conv microbenchmark (scripts/mb_res100.py); run on the GPU box from the repo root
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mb/$tag -- python3 $R/scripts/mb_res100.py > $R/gpurun_out/pmc_mb_$tag.log 2>&1 || echo "pass $tag failed"
done
echo done

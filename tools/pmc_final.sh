#!/bin/bash
# SQ counters of the final pruning kernels (run through gpurun); text summaries under gpurun_out/
for w in c2 c3 c5; do
  bash tools/pmc.sh $w f1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE > gpurun_out/pmc_final_${w}_a.txt 2>&1 || exit 1
  bash tools/pmc.sh $w f2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE > gpurun_out/pmc_final_${w}_b.txt 2>&1 || exit 1
  echo "== $w"; grep -A9 "rt_jit_prune" gpurun_out/pmc_final_${w}_a.txt | head -10; grep -A9 "rt_jit_prune" gpurun_out/pmc_final_${w}_b.txt | head -10
done

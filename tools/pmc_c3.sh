#!/bin/bash
# usage (through gpurun): tools/pmc_c3.sh [workload] -> gpurun_out/pmc_<w>_*.txt
# SQ / LDS / L2 counters of the pruning kernel, a few per pass (separate rocprofv3 runs).
W=${1:-c3}
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" ; do
  tag=$(echo $group | cut -d' ' -f1)
  tools/pmc.sh $W $tag $group > gpurun_out/pmc_${W}_$tag.txt 2>&1 || echo "pass $tag failed"
  cat gpurun_out/pmc_${W}_$tag.txt | grep -v "^$" | head -20
done

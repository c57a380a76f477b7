#!/bin/bash
# PMC passes (counters only) for gauss_fused_kernel over tools/prof_ecc.py
mkdir -p gpurun_out/pmc_g
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr"; do
  bash tools/pmc_one.sh gpurun_out/pmc_g/p "$grp" tools/prof_ecc.py 2>&1 | grep "gauss_fused_kernel<unsigned"
  rm -rf gpurun_out/pmc_g/p
done

#!/bin/bash
# dev: counters of the conv kernels of the default chain (separate --pmc passes, --no-pipeline)
mkdir -p gpurun_out; O=gpurun_out/r05_pmc_conv_new.txt; : > $O
timeout 900 bash tools/pmc_kernel.sh "k_conv12_rs|k_conv5_wp" "" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" >> $O 2>&1
cat $O

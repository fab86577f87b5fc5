#!/bin/bash
# round 6, VERDICT r5 item 1(c): WHERE do the LDS bank conflicts of k_conv12_rs come from?  The dev build's ablations of the role-split kernel (TREXHIP_F12_DBG:
# 8 no production = the producer waves idle, 16 no output transform / V3 transform side of the epilogue, 32 no tap loop, 40 = 8 + 32) under the LDS counters
mkdir -p gpurun_out/r06
export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so
for dbg in 0 8 16 32 40; do
  echo "## TREXHIP_F12_DBG=$dbg"
  TREXHIP_F12_DBG=$dbg bash tools/pmc_kernel.sh "k_conv12_rs" "" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" 2>&1 | grep -v amdgpu
done 2>&1 | tee gpurun_out/r06/lds_conflicts.txt

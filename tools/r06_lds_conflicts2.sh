export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so
for dbg in 4 2 1; do
  echo "## TREXHIP_F12_DBG=$dbg"
  TREXHIP_F12_DBG=$dbg bash tools/pmc_kernel.sh "k_conv12_rs" "" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" 2>&1 | grep -v amdgpu
done

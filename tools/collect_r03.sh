#!/bin/bash
# round-3 evidence on the MI355X box: kernel-trace stats + FETCH / WRITE passes (tools/collect_profiles.sh), matrix-pipe / wait / LDS
# counters of the three convolution kernels of the default chain, and the bench line of the driver's command
#   gpurun --timeout 2400 -- 'bash tools/collect_r03.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
bash $ROOT/tools/collect_profiles.sh r03 > "$OUT/r03_collect.log" 2>&1
bash $ROOT/tools/pmc_kernel.sh "wpre" "" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
     "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r03_pmc_conv.txt" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_rows32|k_ccl_lds|k_gather" "--stages segment" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r03_pmc_detect_kernels.txt" 2>&1
cd $ROOT && python bench.py 2>/dev/null | tail -1 > "$OUT/r03_bench_c4.json"
ls -la "$OUT" | tail -20

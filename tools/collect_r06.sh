#!/bin/bash
# round-6 evidence on the MI355X box: kernel-trace stats + FETCH / WRITE passes of the default bench (tools/collect_profiles.sh), the counters of the two
# convolution kernels and of the detect kernels (C4 and C2 / C3 / C5), the batch-of-one timeline, the bench line of the driver's command
#   gpurun --timeout 3000 -- 'bash tools/collect_r06.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
timeout 1200 bash $ROOT/tools/collect_profiles.sh r06 > "$OUT/r06_collect.log" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_conv12_rs|k_conv12_wpre|k_conv5_wp" "" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
     "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r06_pmc_conv.txt" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_rows32|k_ccl_lds|k_gather" "--stages segment" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r06_pmc_detect_kernels.txt" 2>&1
rm -f "$OUT/r06_pmc_detect_configs.txt"
for cfg in C2 C3 C5; do
  { echo "# detect kernels at $cfg (bench.py --config $cfg --stages segment --force-all, --no-pipeline), separate --pmc passes";
    timeout 600 bash $ROOT/tools/pmc_kernel.sh "k_rows|k_ccl_lds|k_gather|k_rowscan|k_link|k_flatten|k_blobs" "--config $cfg --stages segment --force-all" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; } >> "$OUT/r06_pmc_detect_configs.txt" 2>&1
done
python $ROOT/tools/pmc_detect_json.py "$OUT/r06_pmc_detect_configs.txt" > "$OUT/r06_pmc_detect_configs.json"
cd $ROOT
bash tools/r06_batch1_timeline.sh 2 > "$OUT/r06_batch1_timeline.txt" 2>&1
timeout 900 python bench.py 2>/dev/null > "$OUT/r06_bench_stdout.txt"; tail -1 "$OUT/r06_bench_stdout.txt" > "$OUT/r06_bench_c4.json"
cp gpurun_out/bench_detail.json "$OUT/r06_bench_detail.json" 2>/dev/null; cp gpurun_out/bench_secondary.json "$OUT/r06_bench_secondary.json" 2>/dev/null
timeout 400 bash tools/prof_train.sh r06 > /dev/null 2>&1
ls -la "$OUT" | tail -30

#!/bin/bash
# round-4 evidence on the MI355X box: kernel-trace stats + FETCH / WRITE passes (tools/collect_profiles.sh), matrix-pipe / wait / LDS
# counters of the two convolution kernels of the default chain (conv1 now runs inside conv2: k_conv12_wpre), the detect kernels at C4 and C5,
# the fused-vs-two-kernel A/B and the bench line of the driver's command
#   gpurun --timeout 2400 -- 'bash tools/collect_r04.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
timeout 1200 bash $ROOT/tools/collect_profiles.sh r04 > "$OUT/r04_collect.log" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_conv12_wpre|k_conv5_wpre" "" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
     "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r04_pmc_conv.txt" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_rows32|k_ccl_lds|k_gather" "--stages segment" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r04_pmc_detect_kernels.txt" 2>&1
{ echo "# the same at C5 (4096 x 4096, 256 individuals, 64 frames per launch): every frame's ~7.7 k lines are labelled in LDS since round 4";
  bash $ROOT/tools/pmc_kernel.sh "k_rows32|k_ccl_lds|k_gather|k_rowscan|k_link|k_flatten|k_blobs" "--config C5 --stages segment --force-all" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; } >> "$OUT/r04_pmc_detect_kernels.txt" 2>&1
cd $ROOT
{ echo "# conv1 inside conv2 (k_conv12_wpre, default) against the two-kernel chain k_conv1_wpre + k_conv2_wpre2 (TREXHIP_CONV_GEOM bit 28), tools/time_fused12.py, 25600 / 1000 / 100 crops";
  python tools/time_fused12.py 2>/dev/null; python tools/time_fused12.py 1000 2>/dev/null; python tools/time_fused12.py 100 2>/dev/null;
  if [ -f trex_amd/libtrexhip_dev.so ]; then echo "# dev build (-DTREXHIP_DEV_KNOBS, tools/build_dev.sh): pieces of k_conv12_wpre switched off (tools/f12_ablation.sh; zeroed operands raise the clock: read the differences with care)"; bash tools/f12_ablation.sh 2>/dev/null; fi; } > "$OUT/r04_wpre_ablation.txt" 2>&1
timeout 900 python bench.py 2>/dev/null > "$OUT/r04_bench_stdout.txt"; tail -1 "$OUT/r04_bench_stdout.txt" > "$OUT/r04_bench_c4.json"
cp gpurun_out/bench_detail.json "$OUT/r04_bench_detail.json" 2>/dev/null; cp gpurun_out/bench_secondary.json "$OUT/r04_bench_secondary.json" 2>/dev/null
# the training step (128 samples): kernel trace + three timings
timeout 400 bash tools/prof_train.sh r04 > /dev/null 2>&1
ls -la "$OUT" | tail -24

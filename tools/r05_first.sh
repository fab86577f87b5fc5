#!/bin/bash
# round 5, first GPU call: the SIMD overlap micro-benchmark + FETCH / WRITE passes of the detect kernels at C2, C3, C5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"
cd $ROOT
timeout 300 tools/ubench_simd.bin > "$OUT/r05_ubench_simd.txt" 2>&1
for cfg in C2 C3 C5; do
  { echo "# detect kernels at $cfg (bench.py --config $cfg --stages segment --force-all, --no-pipeline), separate --pmc passes";
    timeout 600 bash $ROOT/tools/pmc_kernel.sh "k_rows|k_ccl_lds|k_gather|k_rowscan|k_link|k_flatten|k_blobs" "--config $cfg --stages segment --force-all" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; } >> "$OUT/r05_pmc_detect_configs.txt" 2>&1
done
cat "$OUT/r05_ubench_simd.txt"
tail -40 "$OUT/r05_pmc_detect_configs.txt"

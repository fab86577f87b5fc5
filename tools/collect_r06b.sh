#!/bin/bash
# round 6, second collection (after k_ccl_band and the two-chunk k_rows32b): the detect kernels at C2 / C3 / C5 again, the bench line of the driver's command, the kernel stats of the traced run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
rm -f "$OUT/r06_pmc_detect_configs.txt"
for cfg in C2 C3 C5; do
  { echo "# detect kernels at $cfg (bench.py --config $cfg --stages segment --force-all, --no-pipeline), separate --pmc passes";
    timeout 600 bash $ROOT/tools/pmc_kernel.sh "k_rows|k_ccl_lds|k_ccl_band|k_gather|k_rowscan|k_link|k_flatten|k_blobs" "--config $cfg --stages segment --force-all" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; } >> "$OUT/r06_pmc_detect_configs.txt" 2>&1
done
python $ROOT/tools/pmc_detect_json.py "$OUT/r06_pmc_detect_configs.txt" > "$OUT/r06_pmc_detect_configs.json"
cp "$OUT/r06_pmc_detect_configs.json" $ROOT/profiles/r06_pmc_detect_configs.json      # bench.py reads the newest one for configs.C2 / C3 / C5.frac
cd $ROOT
timeout 1200 bash tools/collect_profiles.sh r06 > "$OUT/r06_collect.log" 2>&1
cp "$OUT/r06_pmc_summary.json" $ROOT/profiles/r06_pmc_summary.json
timeout 900 python bench.py 2>/dev/null > "$OUT/r06_bench_stdout.txt"; tail -1 "$OUT/r06_bench_stdout.txt" > "$OUT/r06_bench_c4.json"
cp gpurun_out/bench_detail.json "$OUT/r06_bench_detail.json" 2>/dev/null; cp gpurun_out/bench_secondary.json "$OUT/r06_bench_secondary.json" 2>/dev/null
bash tools/r06_batch1_timeline.sh 2 > "$OUT/r06_batch1_timeline.txt" 2>&1
python -c "import __graft_entry__ as e; e.smoke()" > "$OUT/r06_smoke.txt" 2>&1; tail -1 "$OUT/r06_smoke.txt"
ls -la "$OUT" | tail -12

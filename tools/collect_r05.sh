#!/bin/bash
# round-5 evidence on the MI355X box: kernel-trace stats + FETCH / WRITE passes of the default bench (tools/collect_profiles.sh), matrix-pipe / wait /
# LDS / instruction counters of the two convolution kernels of the default chain (k_conv12_rs = conv1 inside conv2, role-split; k_conv5_wpair = conv3, pair by pair),
# the detect kernels at C4, the bench line of the driver's command
#   gpurun --timeout 2400 -- 'bash tools/collect_r05.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
timeout 1200 bash $ROOT/tools/collect_profiles.sh r05 > "$OUT/r05_collect.log" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_conv12_rs|k_conv12_wpre|k_conv5_wp" "" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
     "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r05_pmc_conv.txt" 2>&1
bash $ROOT/tools/pmc_kernel.sh "k_rows32|k_ccl_lds|k_gather" "--stages segment" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" > "$OUT/r05_pmc_detect_kernels.txt" 2>&1
cd $ROOT
timeout 900 python bench.py 2>/dev/null > "$OUT/r05_bench_stdout.txt"; tail -1 "$OUT/r05_bench_stdout.txt" > "$OUT/r05_bench_c4.json"
cp gpurun_out/bench_detail.json "$OUT/r05_bench_detail.json" 2>/dev/null; cp gpurun_out/bench_secondary.json "$OUT/r05_bench_secondary.json" 2>/dev/null
timeout 400 bash tools/prof_train.sh r05 > /dev/null 2>&1
# k_posture & co at C3 (25600 blobs per launch): per-call durations
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -- python $ROOT/tools/time_posture.py 256 > /dev/null 2>&1
python - > "$OUT/r05_posture_kernel_times.txt" <<'PY'
import csv, glob
f = glob.glob("/tmp/kp/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    if "posture" in n or "midline" in n:
        d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# kernel durations (us) of tools/time_posture.py 256 (25600 blobs per launch, three calls; the first call includes first-touch effects)")
for n, v in d.items():
    print(n, [round(x, 1) for x in v])
PY
ls -la "$OUT" | tail -24

#!/bin/bash
# dev: kernel-trace statistics of the training step (tools/time_train.py, 128 samples) -> gpurun_out/prof/<tag>_train_step_kernel_stats.csv
#   gpurun --timeout 600 -- 'bash tools/prof_train.sh r04'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-dev}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptrain
PYTHONPATH=$ROOT timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/ptrain -o t --output-format csv -- python $ROOT/tools/time_train.py --steps 12 --cpu-steps 0 > /tmp/ptrain.log 2>&1
F=$(find /tmp/ptrain -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/${TAG}_train_step_kernel_stats.csv"
cd $ROOT
for i in 1 2 3; do timeout 60 python tools/time_train.py --steps 50 --cpu-steps 0 2>/dev/null | tail -1; done > "$OUT/${TAG}_train_step_time.txt"
cat "$OUT/${TAG}_train_step_time.txt"
head -45 "$OUT/${TAG}_train_step_kernel_stats.csv" | cut -c1-60,150-260

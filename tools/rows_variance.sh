#!/bin/bash
# dev: average duration of the pixel pass over N processes on one box (it is bimodal between processes: ~191 vs ~218 us per 256 frames of 2048^2)
N=${1:-6}
export TMPDIR=/tmp
ROOT=$(pwd)
for i in $(seq $N); do
  rm -rf /tmp/rv; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rv -- python $ROOT/bench.py --stages segment --no-pipeline --no-cpu-baseline --no-secondary --steps 10 ${EXTRA} > /tmp/rv.log 2>&1)
  f=$(find /tmp/rv -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and "k_rows32b" in r[0]: print(f"k_rows32b avg {float(r[3])/1000:7.1f} min {float(r[5])/1000:7.1f} max {float(r[6])/1000:7.1f} us")
PY
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -6

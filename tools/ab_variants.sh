#!/bin/bash
# dev: kernel-trace durations of named kernels for several builds of the library (trex_amd/variants/*.so) on ONE box, two rounds
#   gpurun -- 'bash tools/ab_variants.sh "<kernel regex>" "<bench arguments>"'
PAT=${1:-k_conv5_wpre}
ARGS=${2:---no-pipeline --no-cpu-baseline --no-secondary --steps 6}
ROOT=$(pwd)
cp trex_amd/libtrexhip.so /tmp/keep.so
export TMPDIR=/tmp
for round in 1 2; do for v in trex_amd/variants/*.so; do
  cp $v trex_amd/libtrexhip.so; rm -rf /tmp/abk
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python $ROOT/bench.py $ARGS > /tmp/abk.log 2>&1)
  f=$(find /tmp/abk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$PAT" "$(basename $v)" <<'PY'
import csv, re, sys
f, pat, v = sys.argv[1:4]
for r in csv.reader(open(f)):
    if r and re.search(pat, r[0]):
        print(f"{v:16s} {r[0].split('(')[0][:44]:46s} avg {float(r[3])/1000:8.1f} us  min {float(r[5])/1000:8.1f}")
PY
done; done
cp /tmp/keep.so trex_amd/libtrexhip.so

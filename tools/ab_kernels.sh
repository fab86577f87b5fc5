#!/bin/bash
# dev: kernel-trace durations of selected kernels for the old and the new build of the library on ONE box
#   gpurun -- 'bash tools/ab_kernels.sh "<kernel name regex>" "<bench arguments>"'
PAT=${1:-k_gather}
ARGS=${2:---stages segment --no-cpu-baseline --no-secondary --steps 20}
ROOT=$(pwd)
cp trex_amd/libtrexhip.so /tmp/new.so; cp trex_amd/libtrexhip_old.so /tmp/old.so
export TMPDIR=/tmp
for v in old new old new; do
  cp /tmp/$v.so trex_amd/libtrexhip.so; rm -rf /tmp/abk
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python $ROOT/bench.py $ARGS > /tmp/abk.log 2>&1)
  f=$(find /tmp/abk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$PAT" "$v" <<'PY'
import csv, re, sys
f, pat, v = sys.argv[1:4]
for r in csv.reader(open(f)):
    if r and re.search(pat, r[0]):
        print(f"{v:4s} {r[0].split('(')[0][:48]:50s} calls {r[1]:>5s} avg {float(r[3])/1000:8.1f} us  min {float(r[5])/1000:8.1f}  max {float(r[6])/1000:8.1f}")
PY
done
cp /tmp/new.so trex_amd/libtrexhip.so

#!/bin/bash
# PMC counters of selected kernels (dev tool; separate --pmc passes with --kernel-trace only).
#   tools/pmc_kernel.sh 'regex of kernel names' 'bench.py args' "COUNTER SET 1" "COUNTER SET 2" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PAT="$1"; shift
BARGS="$1"; shift
cd /tmp && export TMPDIR=/tmp
for set in "$@"; do
  rm -rf /tmp/pc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-pipeline --steps 2 --warmup 1 $BARGS > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name '*counter_collection.csv' | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
pat = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if pat.search(n):
        key = (n.split('(')[0].replace('void trexhip::', '')[:44], r['Counter_Name'])
        acc[key].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, v in sorted(acc.items()):
    print(k[0].ljust(46), k[1].ljust(34), '%.4g' % (sum(a for a, _ in v) / len(v)), 'dur_us %.1f' % (sum(b for _, b in v) / len(v) / 1e3))
PY
done

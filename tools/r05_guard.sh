#!/bin/bash
# dev: the C4 step with the range guard tripped on 12..24 crops per step, beside the plain step, alternated; kernel times of one tripped run
mkdir -p gpurun_out; O=gpurun_out/r05_guard.txt; : > $O
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-secondary --steps 20 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('plain  ', j['value'], j['ms_per_step'])" | tee -a $O
  python bench.py --no-cpu-baseline --no-secondary --steps 20 --guard-trip 2>&1 | grep "^# bench_detail" | python -c "
import sys,json
l=sys.stdin.read(); j=json.loads(l.split(':',1)[1]); print('tripped', j['value'], j['ms_per_step'], j.get('range_guard'))" | tee -a $O
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gk -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-pipeline --steps 4 --warmup 2 --guard-trip > /tmp/gk.log 2>&1
f=$(find /tmp/gk -name '*kernel_stats.csv' | head -1); python - "$f" <<'PY' | tee -a $GRAFT_REPO_ROOT/$O
import csv,sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:-float(r['TotalDurationNs']))[:18]:
    print(r['Name'].split('(')[0][-60:].ljust(60), r['Calls'].rjust(5), '%.1f us avg' % (float(r['AverageNs'])/1e3), '%.1f min' % (float(r['MinNs'])/1e3), '%.1f max' % (float(r['MaxNs'])/1e3))
PY

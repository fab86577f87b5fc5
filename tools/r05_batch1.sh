cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/b1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --batch 1 --steps 200 --warmup 20 > /tmp/b1.log 2>&1
tail -1 /tmp/b1.log | cut -c1-200
f=$(find /tmp/b1 -name '*kernel_stats.csv' | head -1); python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'trexhip' in r['Name']]
tot=0
for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs'])):
    print(r['Name'].split('(')[0][-50:].ljust(50), r['Calls'].rjust(6), '%.1f us avg' % (float(r['AverageNs'])/1e3)); tot+=float(r['TotalDurationNs'])
print('sum per step (220 steps) us', tot/220/1e3)
PY

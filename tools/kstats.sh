cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pipeline --steps 5 --warmup 2 $KSTATS_ARGS > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'trexhip' in r['Name']:
        print(r['Name'].split('(')[0][:64].ljust(66), r['Calls'].rjust(4), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(9), r['MinNs'].rjust(9), r['MaxNs'].rjust(9))
PY
grep -h '^{"metric"' /tmp/ks.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_us'])"

#!/bin/bash
# round 6: the kernels of two steps of the default bench (C4, 256 frames) on the GPU's own clock: start, duration, gap (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c4t
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 12 --warmup 4 > /tmp/c4t.log 2>&1
tail -1 /tmp/c4t.log | cut -c1-160
f=$(find /tmp/c4t -name '*kernel_trace.csv' | head -1); python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k0 = max(i for i, r in enumerate(rows) if 'k_conv12_rs' in r['Kernel_Name']) 
# back up three network launches
idx = [i for i, r in enumerate(rows) if 'k_conv12_rs' in r['Kernel_Name']]
k0 = idx[-4]
t0 = int(rows[k0]['Start_Timestamp']); prev_end = {}
print("# start_us  dur_us  gap_on_queue_us  queue  kernel")
for r in rows[k0 - 3:idx[-1]]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', '?')
    name = r['Kernel_Name'].split('(')[0].replace('trexhip::', '')[:56]
    print("%9.1f %8.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end.get(q, s)) / 1e3, q, name))
    prev_end[q] = e
PY

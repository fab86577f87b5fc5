#!/bin/bash
# round 6: what ONE frame per call (TRex's default detect_batch_size) looks like on the GPU's own clock: every kernel of a few steps with its start,
# duration and the gap in front of it (rocprofv3 kernel trace; the library's stage timers off, so the identify chain is the replayed hipGraph)
LANES=${1:-2}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/b1t
rocprofv3 --kernel-trace --output-format csv -d /tmp/b1t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --batch 1 --steps 200 --warmup 20 --lanes $LANES > /tmp/b1t.log 2>&1
tail -1 /tmp/b1t.log | cut -c1-160
f=$(find /tmp/b1t -name '*kernel_trace.csv' | head -1); python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the timed region: skip the first 60 % of the kernels (warm-up, graph capture), then print ~3 steps
k0 = int(len(rows) * 0.8)
t0 = int(rows[k0]['Start_Timestamp']); prev_end = t0
print("# start_us  dur_us  gap_us  stream/queue  kernel")
for r in rows[k0:k0 + 75]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('trexhip::', '')[:60]
    print("%9.1f %7.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), name))
    prev_end = max(prev_end, e)
# averages per kernel over the last 60 %
from collections import defaultdict
d = defaultdict(list)
for r in rows[int(len(rows) * 0.4):]:
    d[r['Kernel_Name'].split('(')[0].replace('trexhip::', '')[:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("# average duration per kernel (us), calls")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s %7.1f %6d" % (k, sum(v) / len(v), len(v)))
PY

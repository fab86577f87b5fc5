#!/bin/bash
# round 6: the library's stage timers with and without a system-scope fence per event record (TREXHIP_TIMER_EVENT_FLAGS=0: the default events of rounds 1-5)
mkdir -p gpurun_out/r06
for rep in 1 2; do for fl in 0x20000000 0; do
echo "## TREXHIP_TIMER_EVENT_FLAGS=$fl"
TREXHIP_TIMER_EVENT_FLAGS=$fl timeout 300 python tools/r06_detect.py C4 256 3:1:0 2>&1 | grep inst
TREXHIP_TIMER_EVENT_FLAGS=$fl timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); s = o.get('stage_us', {}); d = o.get('roofline_detect', {}); print(round(o['value']), 'frames/s', round(o['ms_per_step'], 3), 'ms/step', {k: round(v) for k, v in s.items() if v}, 'serial detect pass', d.get('whole_detect_pass_us'), d.get('whole_detect_pass_frac'), 'pipelined', d.get('pipelined_detect_pass_us'))"
done; done 2>&1 | tee gpurun_out/r06/timer_events.txt

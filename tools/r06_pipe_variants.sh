#!/bin/bash
# round 6: the default bench (C4, 256 frames) under pipeline variants, alternated on one box; frames/s and the stage timers
mkdir -p gpurun_out/r06
run() { echo -n "$1: "; env $2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 $3 2>/dev/null | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); s = o.get('stage_us', {}); print(round(o['value']), 'frames/s', round(o['ms_per_step'], 3), 'ms/step', {k: round(v) for k, v in s.items() if v})"; }
for rep in 1 2; do
run "default" "A=1" ""
run "detect at equal priority" "A=1" "--no-detect-priority"
run "three lanes" "A=1" "--lanes 3"
run "k_head_small for every batch" "TREXHIP_HEAD_SMALL_MAX=100000" ""
done 2>&1 | tee gpurun_out/r06/pipe_variants.txt

#!/bin/bash
mkdir -p gpurun_out/r06
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) | tee gpurun_out/r06/tests_fourth.txt
for lanes in 2 3 4; do for rep in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-secondary --batch 1 --steps 400 --warmup 40 --lanes $lanes 2>/dev/null | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); print('batch 1, lanes $lanes:', round(o['value']), 'frames/s', round(o['ms_per_step'] * 1e3, 1), 'us/step', o.get('stage_us'))"; done; done | tee gpurun_out/r06/batch1_b.txt
bash tools/r06_batch1_timeline.sh 3 > gpurun_out/r06/batch1_timeline_l3_b.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r06/bench_c4_quick_b.txt

#!/bin/bash
# round 6, third GPU call: small-batch head / gather split / guard plan: tests, then TRex's default call shape (one frame per call)
mkdir -p gpurun_out/r06
( timeout 1200 python -m pytest tests/test_cnn_gpu.py tests/test_segment_gpu.py tests/test_bench_shape_gpu.py tests/test_full_size_e2e_gpu.py -x -q -m gpu 2>&1 | tail -15 ) | tee gpurun_out/r06/tests_third.txt
for lanes in 2 3 4; do for rep in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-secondary --batch 1 --steps 400 --warmup 40 --lanes $lanes 2>/dev/null | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); print('batch 1, lanes $lanes:', round(o['value']), 'frames/s', round(o['ms_per_step'] * 1e3, 1), 'us/step', o.get('stage_us'))"; done; done | tee gpurun_out/r06/batch1.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-600 | tee gpurun_out/r06/bench_c4_quick.txt

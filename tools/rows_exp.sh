# dev tool: schedules of the wide pixel pass, alternated to see through box-to-box and run-to-run noise (bits 4096 / 8192 need a -DTREXHIP_DEV_KNOBS build)
run(){ echo -n "ORDER=$1 K=$2: "; if [ -n "$2" ]; then export TREXHIP_ROWS_K=$2; else unset TREXHIP_ROWS_K; fi; TREXHIP_ROWS_ORDER=$1 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --stages segment 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['roofline']['avg_launch_us'], j['roofline']['whole_detect_pass_us'])"; }
for i in 1 2 3; do run 0; run 2048; done
timeout 900 python -m pytest tests/test_segment_gpu.py tests/test_bench_shape_gpu.py tests/test_golden_e2e.py -x -q -m gpu 2>&1 | tail -2

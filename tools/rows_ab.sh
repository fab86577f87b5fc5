# dev: A/B of TREXHIP_ROWS_ORDER[:TREXHIP_ROWS_K] settings on the detect stage (C4), alternated:  bash tools/rows_ab.sh "0 1 1:4" [repeats]
run(){ local o=${1%%:*} k=""; [[ $1 == *:* ]] && k=${1##*:}; echo -n "ORDER=$o K=${k:-8}: "; if [ -n "$k" ]; then export TREXHIP_ROWS_K=$k; else unset TREXHIP_ROWS_K; fi
  TREXHIP_ROWS_ORDER=$o timeout 150 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --stages segment 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j.get('roofline_detect') or j['roofline']; print(r.get('avg_launch_us'), r.get('whole_detect_pass_us'))"; }
for i in $(seq 1 ${2:-2}); do for o in $1; do run $o; done; done

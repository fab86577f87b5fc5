set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"; TAG=r03
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/p_$c; rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $BENCH --steps 3 --warmup 1 > "$OUT/${TAG}_pmc_${c}_run.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    { head -1 "$f"; grep 'trexhip::' "$f" | tail -400; } > "$OUT/${TAG}_pmc_${c}.csv"
done
python $ROOT/tools/summarize_pmc.py "$OUT/${TAG}_pmc_FETCH_SIZE.csv" "$OUT/${TAG}_pmc_WRITE_SIZE.csv" > "$OUT/${TAG}_pmc_summary.json"
python3 -c "
import json; j=json.load(open('$OUT/${TAG}_pmc_summary.json'))
for k,v in j['kernels'].items(): print(k[:44], round(v['hbm_bytes']/1e9,3))"

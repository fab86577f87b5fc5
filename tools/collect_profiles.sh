#!/bin/bash
# Runs on the MI355X box (through gpurun): kernel-trace stats + two separate PMC passes of the default bench command,
# filtered to libtrexhip's kernels, written under gpurun_out/prof/ (copy what should be judged into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02'
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $BENCH --steps 5 --warmup 2 > "$OUT/${TAG}_stats_run.log" 2>&1
f=$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)
{ head -1 "$f"; grep 'trexhip::' "$f"; } > "$OUT/${TAG}_bench_c4_kernel_stats.csv"
grep -h "^{\"metric\"" "$OUT/${TAG}_stats_run.log" > "$OUT/${TAG}_bench_c4_profiled_run.json"
for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/p_$c
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $BENCH --steps 3 --warmup 1 > "$OUT/${TAG}_pmc_${c}_run.log" 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    # keep the rows of the LAST timed step only (one dispatch per kernel keeps the file small): last 400 trexhip rows (the bench ends with ~100 detect-pass dispatches of its roofline_detect measurements)
    { head -1 "$f"; grep 'trexhip::' "$f" | tail -400; } > "$OUT/${TAG}_pmc_${c}.csv"
done
python $ROOT/tools/summarize_pmc.py "$OUT/${TAG}_pmc_FETCH_SIZE.csv" "$OUT/${TAG}_pmc_WRITE_SIZE.csv" > "$OUT/${TAG}_pmc_summary.json"
ls -la "$OUT"

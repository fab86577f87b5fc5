cd $GRAFT_REPO_ROOT
cp trex_amd/libtrexhip.so /tmp/new.so; cp trex_amd/libtrexhip_old.so /tmp/old.so
export TMPDIR=/tmp
for v in old new old new; do
  cp /tmp/$v.so trex_amd/libtrexhip.so; rm -rf /tmp/kp
  (cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python $GRAFT_REPO_ROOT/tools/time_posture.py 256 > /tmp/kp.log 2>&1)
  f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and ("posture" in r[0] or "midline" in r[0]):
        print(sys.argv[2], r[0].split("(")[0][:40].ljust(42), "calls", r[1], "avg %.1f us" % (float(r[3]) / 1e3), "min %.1f" % (float(r[5]) / 1e3))
PY
done
cp /tmp/new.so trex_amd/libtrexhip.so

export TMPDIR=/tmp
for gb in 1280 1600 2048 2560 3200 6400; do
  rm -rf /tmp/abk; (cd /tmp && TREXHIP_GATHER_BLOCKS=$gb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python /root/repo/bench.py --stages segment --no-pipeline --no-cpu-baseline --no-secondary --steps 20 > /tmp/abk.log 2>&1)
  f=$(find /tmp/abk -name "*kernel_stats.csv" | head -1); echo -n "blocks $gb: "; grep k_gather $f | sed 's/(.*)"/"/' | awk -F'",' '{print $2}' | cut -d, -f3
done

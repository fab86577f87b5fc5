#!/bin/bash
# dev: A/B of two builds of the library on ONE box, identity network only (tools/time_fused12.py's "fused" row), alternated
#   cp trex_amd/libtrexhip.so trex_amd/libtrexhip_old.so (before the change), rebuild, gpurun -- 'bash tools/ab_identify.sh [n] [reps]'
N=${1:-25600}; R=${2:-3}
for it in $(seq $R); do
  for v in old new; do
    if [ $v = old ]; then export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_old.so; else export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip.so; fi
    echo "$v $(python tools/time_fused12.py $N 2>/dev/null | grep fused)"
  done
done

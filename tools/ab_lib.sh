#!/bin/bash
# dev: A/B of two builds of the library within ONE box (box-to-box variance exceeds most kernel-level gains)
#   cp trex_amd/libtrexhip.so trex_amd/libtrexhip_old.so   (before the change), rebuild, then
#   gpurun -- 'bash tools/ab_lib.sh "<bench arguments>" [repetitions]'
ARGS=${1:---no-cpu-baseline --no-secondary --steps 20}
N=${2:-3}
cp trex_amd/libtrexhip.so /tmp/new.so; cp trex_amd/libtrexhip_old.so /tmp/old.so
for it in $(seq $N); do
  for v in old new; do cp /tmp/$v.so trex_amd/libtrexhip.so; echo -n "$v "; python bench.py $ARGS 2>&1 | tail -1 | cut -c1-40,80-200; done
done
cp /tmp/new.so trex_amd/libtrexhip.so

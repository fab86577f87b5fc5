#!/bin/bash
# dev: the identity network compiled with and without the packed fp32 vector instructions, alternated on one box (the product build disables them)
mkdir -p gpurun_out; O=gpurun_out/r05_pk.txt; : > $O
for i in 1 2; do
for L in libtrexhip_dev.so libtrexhip_devpk.so; do
  echo "== $L" | tee -a $O
  TREXHIP_LIB_PATH=$PWD/trex_amd/$L timeout 300 python tools/time_fused12.py 2>&1 | grep "fused\|max" | tee -a $O
done; done

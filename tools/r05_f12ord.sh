#!/bin/bash
# dev: the role-split conv1 + conv2 kernel with pair-major taps (ORD 1, the default) against kernel-row-major taps (TREXHIP_F12_DBG=230), alternated;
# bit-identity against the two-kernel chain is time_fused12.py's own check; stage stamps of both (128 / 231)
mkdir -p gpurun_out; O=gpurun_out/r05_f12ord.txt; : > $O
export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so
for i in 1 2; do
  echo "== ORD 1" | tee -a $O; timeout 300 python tools/time_fused12.py 2>&1 | grep "fused\|max" | tee -a $O
  echo "== ORD 0" | tee -a $O; TREXHIP_F12_DBG=230 timeout 300 python tools/time_fused12.py 2>&1 | grep "fused\|max" | tee -a $O
done
echo "== stamps ORD 1" | tee -a $O; TREXHIP_F12_DBG=128 timeout 300 python tools/f12rs_stamps.py 2>&1 | tail -2 | tee -a $O
echo "== stamps ORD 0" | tee -a $O; TREXHIP_F12_DBG=231 timeout 300 python tools/f12rs_stamps.py 2>&1 | tail -2 | tee -a $O

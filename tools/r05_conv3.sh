#!/bin/bash
# dev: the pair-major conv3 (k_conv5_wpair) against the previous build of the library (trex_amd/libtrexhip_old.so) on one box: probabilities of the
# same crops (sizes with partial passes), CONV3 stage time alternated old / new, then the dev build's ablations and weight leads
mkdir -p gpurun_out; O=gpurun_out/check_conv3.txt; : > $O
TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_old.so timeout 300 python tools/check_conv3.py 0 save:/tmp/c3ref.npz 2>&1 | grep variant | sed 's/^/old  /' | tee -a $O
timeout 300 python tools/check_conv3.py 0 /tmp/c3ref.npz 2>&1 | grep variant | sed 's/^/new  /' | tee -a $O
TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_old.so timeout 300 python tools/check_conv3.py 0 2>&1 | grep variant | sed 's/^/old  /' | tee -a $O
timeout 300 python tools/check_conv3.py 0 /tmp/c3ref.npz 2>&1 | grep variant | sed 's/^/new  /' | tee -a $O
TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so timeout 600 python tools/check_conv3.py ${1:-0,9,10,11,1,2,3,7,15} /tmp/c3ref.npz 2>&1 | grep variant | sed 's/^/dev  /' | tee -a $O

#!/bin/bash
# dev: variants of the pair-major conv3 in the dev build (TREXHIP_CONV_GEOM bits 24..27, tools/check_conv3.py): probabilities against variant 0, CONV3 stage time
mkdir -p gpurun_out
TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so timeout 600 python tools/check_conv3.py $1 2>&1 | grep variant | tee gpurun_out/check_conv3_dev.txt

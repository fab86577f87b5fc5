#!/bin/bash
# round 6, second GPU call: where k_ccl_lds spends its time (dev build stamps), fused against separate gather at few frames per launch, the range guard tests
mkdir -p gpurun_out/r06
for a in "256 C4" "1 C4" "64 C5" "256 C2"; do TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so timeout 300 python tools/ccl_stamps.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06/ccl_stamps.txt
for fg in 1 0; do echo "## TREXHIP_FUSE_GATHER=$fg"; for a in "C4 1" "C4 16" "C5 64" "C5 16" "C2 64"; do TREXHIP_FUSE_GATHER=$fg timeout 300 python tools/r06_detect.py $a 3:1:0 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r06/fuse_gather.txt
( timeout 900 python -m pytest tests/test_cnn_gpu.py -x -q -m gpu 2>&1 | tail -3 ) | tee gpurun_out/r06/tests_cnn.txt

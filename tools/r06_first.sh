#!/bin/bash
# round 6, first GPU call: correctness of the k_ccl_lds instances, then the detect variants and the V3-residency question
mkdir -p gpurun_out/r06
( timeout 900 python -m pytest tests/test_segment_gpu.py tests/test_rethreshold_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r06/tests_segment.txt
cat gpurun_out/r06/tests_segment.txt
for inst in 1 2 5; do ( TREXHIP_CCL_INST=$inst timeout 600 python -m pytest tests/test_segment_gpu.py -x -q -m gpu 2>&1 | tail -2 ) | tee -a gpurun_out/r06/tests_segment.txt; done
timeout 600 python tools/r06_detect.py C4 256 2>&1 | tee gpurun_out/r06/detect_c4.txt
timeout 600 python tools/r06_detect.py C2 256 3:1:0 2:1:0 1:1:0 5:1:0 0:1:0 1:2:1 1:2:2 2>&1 | tee gpurun_out/r06/detect_c2.txt
timeout 600 python tools/r06_detect.py C2 1024 3:1:0 2:1:0 1:1:0 5:1:0 1:2:2 2>&1 | tee gpurun_out/r06/detect_c2_1024.txt
timeout 600 python tools/r06_detect.py C5 64 3:1:0 3:2:1 3:2:2 3:4:1 3:4:2 2>&1 | tee gpurun_out/r06/detect_c5.txt
timeout 600 python tools/r06_detect.py C4 1 3:1:0 2:1:0 4:1:0 1:1:0 0:1:0 2>&1 | tee gpurun_out/r06/detect_c4_b1.txt
timeout 900 python tools/r06_v3_resident.py 2>&1 | tee gpurun_out/r06/v3_resident.txt

#!/bin/bash
# the randomised parity sweeps of every stage on the current build (tallies -> profiles/rNN_fuzz_sweeps.txt)
#   gpurun --timeout 2400 -- 'bash tools/fuzz_all.sh > gpurun_out/fuzz_all.txt 2>&1'
export PYTHONPATH=.:tests
echo "# detect 3000 511"; timeout 900 python tools/fuzz_detect.py 3000 511 2>&1 | tail -1
echo "# re-threshold 800 512"; timeout 600 python tools/fuzz_rethreshold.py 800 512 2>&1 | tail -1
echo "# posture 400 513"; timeout 600 python tools/fuzz_posture.py 400 513 2>&1 | tail -1
echo "# crops 150 514"; timeout 400 python tools/fuzz_crops.py 150 514 2>&1 | tail -1
echo "# split 300 515"; timeout 400 python tools/fuzz_split.py 300 515 2>&1 | tail -1
echo "# cnn 120 516"; timeout 400 python tools/fuzz_cnn.py 120 516 2>&1 | tail -1
echo "# train 120 517"; timeout 600 python tools/fuzz_train.py 120 517 2>&1 | tail -1

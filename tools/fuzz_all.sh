#!/bin/bash
# the randomised parity sweeps of every stage on the current build (tallies -> profiles/rNN_fuzz_sweeps.txt)
#   gpurun --timeout 2400 -- 'bash tools/fuzz_all.sh > gpurun_out/fuzz_all.txt 2>&1'
export PYTHONPATH=.:tests
echo "# detect 4000 411"; python tools/fuzz_detect.py 4000 411 2>&1 | tail -1
echo "# re-threshold 1000 412"; python tools/fuzz_rethreshold.py 1000 412 2>&1 | tail -1
echo "# posture 500 413"; python tools/fuzz_posture.py 500 413 2>&1 | tail -1
echo "# crops 150 414"; python tools/fuzz_crops.py 150 414 2>&1 | tail -1
echo "# split 400 415"; python tools/fuzz_split.py 400 415 2>&1 | tail -1
echo "# cnn 160 416"; python tools/fuzz_cnn.py 160 416 2>&1 | tail -1
echo "# train 100 417"; python tools/fuzz_train.py 100 417 2>&1 | tail -1

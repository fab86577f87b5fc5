#!/bin/bash
# round 6: the detect fuzzer (batches of 1-16 frames against the oracle) with the new labelling paths forced: bands, the small instances with their retry, gather fused / by its own launch, fetch by DMA copies
export PYTHONPATH=.:tests
for v in "TREXHIP_CCL_BANDS=3" "TREXHIP_CCL_BANDS=8" "TREXHIP_CCL_INST=1" "TREXHIP_CCL_INST=2" "TREXHIP_FUSE_GATHER=1" "TREXHIP_FUSE_GATHER=0" "TREXHIP_EXPORT=0" "TREXHIP_ROWS_ORDER=8"; do
  echo "# detect 700 $v"; env $v timeout 600 python tools/fuzz_detect.py 700 601 2>&1 | tail -1
done
echo "# re-threshold 300 TREXHIP_CCL_BANDS=4"; TREXHIP_CCL_BANDS=4 timeout 600 python tools/fuzz_rethreshold.py 300 602 2>&1 | tail -1
echo "# split 150 TREXHIP_CCL_BANDS=4"; TREXHIP_CCL_BANDS=4 timeout 400 python tools/fuzz_split.py 150 603 2>&1 | tail -1
echo "# posture 150 outline_approximate default"; timeout 600 python tools/fuzz_posture.py 150 604 2>&1 | tail -1

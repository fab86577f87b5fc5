#!/bin/bash
# round 6: the kernels behind the two convolutions (fc1, head, guard chain) of an identify call over 25600 crops, average durations under rocprofv3, for variants of fc1's tile
mkdir -p gpurun_out/r06
cat > /tmp/idn.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from trex_amd import capi, weights
st = weights.synthetic_state(100, 31)
seg = capi.Segmenter(capi.default_params(64, 64, max_batch=1), stream=None); seg.load_weights(weights.pack_blob(st, 100))
N = 25600
crops = torch.from_numpy(np.tile(weights.synthetic_crops(100, 3), (N // 100, 1, 1, 1))).cuda()
probs = torch.zeros((N, 100), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
for _ in range(8):
    seg.identify_device(crops.data_ptr(), N, probs.data_ptr())
seg.synchronize()
print("checksum", float(probs.double().sum()), float(probs[::997].double().abs().sum()))
PY
for mt in 4 2 1; do
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/idn
  TREXHIP_FC1_MT=$mt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/idn -- python /tmp/idn.py 2>/dev/null | grep checksum
  f=$(find /tmp/idn -name '*kernel_stats.csv' | head -1)
  echo "## TREXHIP_FC1_MT=$mt"; python - "$f" <<'PY'
import csv, sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r['TotalDurationNs'])):
    if 'trexhip' in r['Name']: print(r['Name'].split('(')[0][-58:].ljust(58), r['Calls'].rjust(4), '%9.1f us avg  min %9.1f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r06/tail_kernels.txt

#!/bin/bash
# round 5: the role-split conv1+conv2 kernel (default) against round 4's two-workgroup kernel (TREXHIP_CONV_GEOM bit 29) and the two-kernel chain (bit 28)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"
cd $ROOT
# small batch first, under a short timeout: a barrier mismatch would hang
timeout 60 python tools/time_fused12.py 100 2>&1 | tail -4 || { echo "HANG or failure at 100 crops"; exit 1; }
timeout 60 python tools/time_fused12.py 1000 2>&1 | tail -4 || { echo "HANG or failure at 1000 crops"; exit 1; }
timeout 120 python tools/time_fused12.py 2>&1 | tail -4
echo "# round 4 kernel (bit 29)"
timeout 120 env TREXHIP_F12_OLD=1 python tools/time_fused12.py 2>&1 | tail -3

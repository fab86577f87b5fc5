#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"
cd $ROOT
export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so
{
echo "# phase stamps of k_conv12_wpre, vector phases prio 3 / tap loop 0 (round 5 default)"
TREXHIP_F12_DBG=128 python tools/f12_stamps.py 2>/dev/null | tail -2
echo "# the same with the priorities of rounds 3-4 (vector 0 / tap 3)"
TREXHIP_F12_DBG=129 python tools/f12_stamps.py 2>/dev/null | tail -2
echo "# ONE workgroup per CU (TREXHIP_F12_WGS=1): stamps, then time"
TREXHIP_F12_WGS=1 TREXHIP_F12_DBG=128 python tools/f12_stamps.py 2>/dev/null | tail -2
TREXHIP_F12_WGS=1 python tools/time_fused12.py 2>/dev/null | grep fused
echo "# ablations under the new priorities"
F12="0 8 16 32 40 64" bash tools/f12_ablation.sh
} > "$OUT/r05_f12_stamps.txt" 2>&1
cat "$OUT/r05_f12_stamps.txt"

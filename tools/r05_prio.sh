#!/bin/bash
# round 5: priorities of the fused conv1+conv2 kernel's phases (dev build: TREXHIP_F12_DBG 100 = vector 0 / tap 3 (rounds 3-4), 101 = 1 / 0, 102 = 0 / 0, 103 = 3 / 1; default = 3 / 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"
cd $ROOT
{
echo "# k_conv12_wpre: s_setprio of (vector phases / tap loop), tools/time_fused12.py 25600 crops, dev build, one box"
for rep in 1 2; do
for d in 0 100 101 102 103; do echo "f12 variant $d: $(TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so TREXHIP_F12_DBG=$d python tools/time_fused12.py 2>/dev/null | grep -E 'fused|max' | tr '\n' ' ')"; done
done
} > "$OUT/r05_prio_ablation.txt" 2>&1
cat "$OUT/r05_prio_ablation.txt"
timeout 600 python -m pytest tests/test_cnn_gpu.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-1500

# dev tool (needs tools/build_dev.sh): where the time of the fused conv1+conv2 kernel goes.  TREXHIP_F12_DBG: 1 no crop loads, 2 no conv1 MFMAs,
# 4 no P2 transform, 8 no production at all, 16 no epilogue, 32 no tap loop (combinations: 24, 40, 56)
export TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so
for d in ${F12:-0 1 2 4 8 16 24 32 40 56}; do echo "f12 dbg $d: $(TREXHIP_F12_DBG=$d python tools/time_fused12.py 2>/dev/null | grep fused | head -1)"; done

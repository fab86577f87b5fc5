# dev tool (needs a -DTREXHIP_DEV_KNOBS build): ablations of the default chain's conv2 (k_conv2_wpre, TREXHIP_CONV_GEOM bits 16..21:
# 1 no staging, 2 no epilogue, 4 no weight loads, 8 no A reads, 16 staging loads from one hot row, 32 no V3 transform; 48 / 49: weights 3 / 5 taps ahead)
# and conv3 (k_conv5_wpre, bits 24..27: the same, 8 = hot row, 9 / 10 = weights 3 / 5 taps ahead)
for d in ${W2:-0 1 2 3 7 15 32 16 48 49}; do echo "conv2 dbg $d: $(TREXHIP_CONV_GEOM=$((d*65536)) python tools/time_wino.py 2>/dev/null | grep -E 'CONV2|CNN_ALL' | tr '\n' ' ')"; done
for d in ${W3:-0 1 2 3 7 15 8 9 10}; do echo "conv3 dbg $d: $(TREXHIP_CONV_GEOM=$((d*16777216)) python tools/time_wino.py 2>/dev/null | grep -E 'CONV3|CNN_ALL' | tr '\n' ' ')"; done

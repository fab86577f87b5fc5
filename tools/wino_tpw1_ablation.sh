# dev tool (needs -DTREXHIP_DEV_KNOBS): ablations of conv3's one-M-tile-per-wave Winograd variant (TREXHIP_CONV_GEOM bit 10) and of the default
for base in 1024 0; do for d in 0 1 2 3 4 7 15; do echo -n "base $base dbg $d: "; TREXHIP_CONV_GEOM=$((base + d*4096)) python tools/time_wino.py 2>/dev/null | grep CONV3; done; done

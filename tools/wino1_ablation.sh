for d in 0 1 2 3 7 15 4 8; do TREXHIP_CONV_GEOM=$((512 + d*4096)) python tools/time_wino.py 2>/dev/null | grep CONV2; done
TREXHIP_CONV_GEOM=0 python tools/time_wino.py 2>/dev/null | grep CONV2

mkdir -p gpurun_out
TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so timeout 600 python tools/check_conv3.py $1 2>&1 | grep variant | tee gpurun_out/check_conv3_dev.txt

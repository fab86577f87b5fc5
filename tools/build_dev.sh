#!/bin/bash
# dev: a -DTREXHIP_DEV_KNOBS build of the library next to the product one (trex_amd/libtrexhip_dev.so; chosen with TREXHIP_LIB_PATH=$PWD/trex_amd/libtrexhip_dev.so by the dev
# tools only -- capi.py never loads it by itself).  cnn.hip / segment.hip / capi.hip carry knobs; the other objects are shared with the product build.
cd "$(dirname "$0")/../trex_amd/csrc" || exit 1
make -s -j8 || exit 1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -DTREXHIP_DEV_KNOBS"
/opt/rocm/bin/hipcc $F -Xclang -target-feature -Xclang -packed-fp32-ops -c cnn.hip -o /tmp/cnn_dev.o &
/opt/rocm/bin/hipcc $F -c segment.hip -o /tmp/segment_dev.o &
/opt/rocm/bin/hipcc $F -c capi.hip -o /tmp/capi_dev.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtrexhip_dev.so /tmp/capi_dev.o /tmp/segment_dev.o /tmp/cnn_dev.o crops.o morph.o posture.o midline.o split.o upload.o comm.o pack.o train.o hostcvt.o pvfile.o -lpthread -ldl
ls -la ../libtrexhip_dev.so

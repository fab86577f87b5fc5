#!/bin/bash
# dev tool: ISA of one kernel of cnn.hip (or another source) + where its scratch traffic sits relative to the MFMA loop
#   tools/isa_kernel.sh <mangled-name-prefix> [source.hip]
SRC=${2:-cnn.hip}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-everything -S --cuda-device-only -c /root/repo/trex_amd/csrc/$SRC -I/root/repo/trex_amd/csrc -o /tmp/k.s || exit 1
L0=$(grep -n "^$1.*:" /tmp/k.s | head -1 | cut -d: -f1)
L1=$(grep -n "\.amdhsa_kernel $1" /tmp/k.s | head -1 | cut -d: -f1)
sed -n "${L0},${L1}p" /tmp/k.s > /tmp/kernel.s
F=$(grep -n "v_mfma" /tmp/kernel.s | head -1 | cut -d: -f1); L=$(grep -n "v_mfma" /tmp/kernel.s | tail -1 | cut -d: -f1)
echo "lines $(wc -l < /tmp/kernel.s)  mfma $(grep -c v_mfma /tmp/kernel.s)  first/last mfma line $F/$L"
echo "scratch ops: total $(grep -c scratch_ /tmp/kernel.s)  inside mfma range $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c scratch_)"
echo "inside mfma range: valu $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c '^\sv_' ) ds $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c '^\sds_') vmem $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c '^\sglobal_') salu $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c '^\ss_') waitcnt $(sed -n "${F},${L}p" /tmp/kernel.s | grep -c 's_waitcnt')"

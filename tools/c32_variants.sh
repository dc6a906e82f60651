#!/bin/bash
# Lab: the persistent 32 -> 32 channel Winograd kernel (csrc/conv_wino_c32.h) built with parts switched off (MTD_C32_SKIP bits:
# 1 no MFMAs, 2 no input transform, 4 no output transform, 8 no patch loads), each timed on one whole-slice layer by
# tools/c32_conv_time.py.  The results are wrong; the times say which part the block time is made of.  Usage (on the box):
#   bash tools/c32_variants.sh        (C32_CFGS="-DMTD_C32_SKIP=1;-DMTD_C32_SKIP=7;..." to choose: ';' between builds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OBJS=$(ls mtd-gan_amd/csrc/build/*.o | grep -v conv_winograd.o)
IFS=';' read -ra CFGS <<< "${C32_CFGS:--DMTD_C32_SKIP=0;-DMTD_C32_SKIP=1;-DMTD_C32_SKIP=2;-DMTD_C32_SKIP=4;-DMTD_C32_SKIP=8;-DMTD_C32_SKIP=7;-DMTD_C32_SKIP=14;-DMTD_C32_SKIP=15}"
for cfg in "${CFGS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $cfg -c mtd-gan_amd/csrc/conv_winograd.hip -o mtd-gan_amd/csrc/build/conv_winograd.o 2>/dev/null || { echo "compile failed: $cfg"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mtd-gan_amd/libmtdgan_hip.so $OBJS mtd-gan_amd/csrc/build/conv_winograd.o
  echo "== $cfg"
  MTD_NO_BUILD=1 timeout -k 10 100 python tools/c32_conv_time.py ${C32_SIZE:-8 512 512} 2>&1 | grep "us "
done

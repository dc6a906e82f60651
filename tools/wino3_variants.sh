#!/bin/bash
# Lab: the split-bf16 Winograd kernel with parts switched off (W3_SKIP, conv_winograd_split.h).  Builds the variants HERE (no GPU
# needed); on the GPU box: for v in 0 1 2 4 8 16 ...; do cp tools/w3lab/lib_$v.so mtd-gan_amd/libmtdgan_hip_lab.so; MTD_LAB=1 W3_ONLY=1 python tools/wino3_probe.py; done
cd "$(dirname "$0")/.."
mkdir -p tools/w3lab
for v in "$@"; do
  touch mtd-gan_amd/csrc/conv_winograd.hip
  MTD_LAB_BUILD=1 MTD_LAB_FLAGS="-DW3_SKIP=$v" python mtd-gan_amd/_build.py > /dev/null 2>&1 || { echo "build $v failed"; exit 1; }
  cp mtd-gan_amd/libmtdgan_hip_lab.so tools/w3lab/lib_$v.so
  echo "built variant $v"
done
rm -f mtd-gan_amd/libmtdgan_hip_lab.so

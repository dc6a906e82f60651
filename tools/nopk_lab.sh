#!/bin/bash
# Lab: the library's translation units rebuilt WITHOUT packed-fp32 vector instructions (v_pk_fma_f32 & co. cost 7 clocks for two
# floats per lane where two plain instructions cost 6.2: tools/valu_rate_probe.hip), one unit at a time and all together, each
# timed by the full training step.  Usage (on the box): bash tools/nopk_lab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=mtd-gan_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
step() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-generator --no-inference --no-engine-api 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mtd-gan_amd/libmtdgan_hip.so $B/*.o; }
echo "as built: $(step) $(step)"
for u in ${UNITS:-conv_wgrad conv_c32_bwd conv_igemm resfft resfft4 conv_winograd}; do
  cp $B/$u.o /tmp/$u.keep.o
  /opt/rocm/bin/hipcc $FLAGS $NOPK -c mtd-gan_amd/csrc/$u.hip -o $B/$u.o 2>/dev/null || { echo "compile failed: $u"; cp /tmp/$u.keep.o $B/$u.o; continue; }
  link
  echo "$u without packed fp32: $(step) $(step)"
  cp /tmp/$u.keep.o $B/$u.o
done
link

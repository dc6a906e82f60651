#!/bin/bash
# Lab: the whole-slice spectral kernels (csrc/resfft_any.hip) built with different compile-time knobs on the GPU box, each timed
# by the inference512 workload's roofline pass.  Usage (on the box): bash tools/any_variants.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
OBJS=$(ls mtd-gan_amd/csrc/build/*.o | grep -v resfft_any.o)
for cfg in ${ANY_CFGS:-"" "-DMTD_ANY_FFTV=1" "-DMTD_ANY_SKIP=1" "-DMTD_ANY_SKIP=2" "-DMTD_ANY_SKIP=4" "-DMTD_ANY_SKIP=7"}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $cfg -c mtd-gan_amd/csrc/resfft_any.hip -o mtd-gan_amd/csrc/build/resfft_any.o 2>/dev/null || { echo "compile failed: $cfg"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mtd-gan_amd/libmtdgan_hip.so $OBJS mtd-gan_amd/csrc/build/resfft_any.o
  timeout -k 10 200 python bench.py --workload inference512 --no-cpu-baseline --steps 10 --warmup 2 > $O/any_var.json 2> $O/any_var.err || { echo "bench failed: $cfg"; tail -3 $O/any_var.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
z = json.load(open("gpurun_out/any_var.json"))
r = z["roofline"]
o = dict(r.get("other_mfma_kernels") or r.get("other_kernels") or {})
o[r["kernel"]] = {"ms_per_step": r["share_of_step_gpu_ms"]}
print(f"{sys.argv[1]!r:50s} step {z['ms_per_step']:.2f} ms  rows {o['rfft_rows_any_kernel']['ms_per_step']:.2f}  mix {o['spec_mix_any_kernel']['ms_per_step']:.2f}  rows back {o['irfft_rows_any_kernel']['ms_per_step']:.2f}", flush=True)
PY
done

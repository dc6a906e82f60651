#!/bin/bash
# Copies the summaries of the last tools/measure.sh run from gpurun_out/ (scratch) to profiles/ (tracked)
# and rewrites profiles/pmc_manifest.json with the kernel-source hash the PMC passes were taken at.  Run after a gpurun call
# of the script, with the same kernel sources checked out.   usage: tools/install_profiles.sh r6 [nogit]
set -e
cd "$(dirname "$0")/.."
R=${1:-r6}
O=gpurun_out
[ -f $O/bench_full.json ] && [ "$2" != "nogit" ] && cp $O/bench_full.json profiles/${R}_full_step_bench.json
cp $O/prof_full/full_results_kernel_stats.csv profiles/${R}_full_step_kernel_stats_single_stream.csv
cp $O/prof_gen/gen_results_kernel_stats.csv profiles/${R}_generator_kernel_stats_single_stream.csv
cp $O/prof_inf/inf_results_kernel_stats.csv profiles/${R}_inference512_kernel_stats.csv
for wl in full_step generator inference512; do
  for k in fetch write mfma_util; do cp $O/pmc_${wl}_$k.csv profiles/${R}_pmc_${wl}_$k.csv; done
done
python - "$R" <<'PY'
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
R = sys.argv[1]
h_run = open("gpurun_out/pmc_source_hash.txt").read().strip()
h_now = bench._kernel_source_hash()
assert h_run == h_now, f"PMC passes were taken at source hash {h_run}, the tree is at {h_now}"
try:
    commit = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    commit = "unknown (no .git on the GPU box; the builder's install_profiles.sh run records it)"
try:
    lab = open("gpurun_out/pmc_lab_build.txt").read().strip()      # "<mtd_lab_build()> <path of the library the passes ran on>"
except OSError:
    lab = "unrecorded"
assert lab == "unrecorded" or lab.startswith("0 "), f"PMC passes ran on a lab build: {lab}"
cmd = f"tools/measure.sh {R} (rocprofv3 --pmc <counter(s)> --kernel-trace, one pass per counter set, eager single-stream launches)"
man = {wl: {"files": {"fetch": f"{R}_pmc_{wl}_fetch.csv", "write": f"{R}_pmc_{wl}_write.csv", "mfma": f"{R}_pmc_{wl}_mfma_util.csv"},
            "commit": commit, "source_hash": h_now, "command": cmd, "library": lab} for wl in ("full_step", "generator", "inference512")}
json.dump(man, open("profiles/pmc_manifest.json", "w"), indent=1)
print("profiles/ updated at", commit[:10], h_now)
PY

#!/bin/bash
# GPU-box measurement pass (usage: tools/measure.sh r6; SKIP_PMC=1 for the tables only): the bench line, the single-stream rocprofv3 kernel tables the roofline figures are compared with, and the
# PMC passes of the three bench workloads (separate runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE, WRITE_SIZE and the MFMA
# busy counters each in a pass of its own, kernel trace only; eager single-stream launches so that every dispatch is attributed).
# Results under gpurun_out/; tools/install_profiles.sh <round> copies them to profiles/ with the kernel-source hash they belong to.
# Every pass runs the SHIPPED library (MTD_LAB_LIB=0): MTD_LAB=1 only makes the Python-level single-stream / eager switches live.
R=${1:-r6}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
rm -rf $O/prof_full $O/prof_gen $O/prof_inf
MTD_LAB=1 MTD_LAB_LIB=0 MTD_LIST=0 MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 bench.py --steps 5 --warmup 2 $NOX > $O/prof_full.log 2>&1 || { echo "rocprof full failed"; tail -5 $O/prof_full.log; exit 1; }
MTD_LAB=1 MTD_LAB_LIB=0 MTD_NO_SIDE_STREAMS=1 MTD_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o gen -- python3 bench.py --workload generator --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/prof_gen.log 2>&1 || { echo "rocprof gen failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_inf -o inf -- python3 bench.py --workload inference512 --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $O/prof_inf.log 2>&1 || { echo "rocprof inference failed"; exit 1; }
find $O/prof_gen $O/prof_full $O/prof_inf -name "*.db" | while read f; do
  case $f in *gen*) n=13;; *) n=7;; esac
  python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps $n
done
find $O/prof_gen $O/prof_full $O/prof_inf -name "*.db" -size +30M -delete
echo "kernel tables done"
if [ -z "$SKIP_PMC" ]; then
  export MTD_LAB=1 MTD_LAB_LIB=0 MTD_GRAPH=0 MTD_NO_SIDE_STREAMS=1 MTD_LIST=0
  for wl in full_step generator inference512; do
    for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
      name=${pass%%:*}; ctrs=${pass#*:}
      rm -rf $O/pmc_${wl}_$name
      timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_${wl}_$name -- python3 bench.py --workload $wl --steps 3 --warmup 1 $NOX > $O/pmc_${wl}_$name.log 2>&1 || { echo "pmc pass $wl $name failed"; tail -5 $O/pmc_${wl}_$name.log; exit 1; }
      python tools/pmc_summary.py $O/pmc_${wl}_$name $O/pmc_${wl}_$name.csv
      find $O/pmc_${wl}_$name -type f -size +4M -delete
      echo "pmc $wl $name done"
    done
    python tools/pmc_mfma_util.py $O/pmc_${wl}_mfma.csv $O/pmc_${wl}_mfma_util.csv
  done
  python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
from mtd_gan_amd import _lib
open("gpurun_out/pmc_source_hash.txt", "w").write(bench._kernel_source_hash())
open("gpurun_out/pmc_lab_build.txt", "w").write("%d %s" % (_lib.lib().mtd_lab_build(), _lib.LIB_PATH))
PY
  unset MTD_LAB MTD_LAB_LIB MTD_GRAPH MTD_NO_SIDE_STREAMS MTD_LIST
  # the bench line with profiles/ on THIS box holding PMC passes of THESE kernel sources (its roofline objects quote traffic and MFMA
  # utilisation only from passes whose source hash matches)
  bash tools/install_profiles.sh $R nogit || exit 1
fi
timeout -k 10 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { echo "bench failed"; tail -20 $O/bench_full.err; exit 1; }
cut -c1-400 $O/bench_full.json; echo

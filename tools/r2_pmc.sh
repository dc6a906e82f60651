#!/bin/bash
# PMC passes of both bench workloads (separate runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE, WRITE_SIZE and the MFMA
# busy counters each in a pass of its own, kernel trace only).  Eager single-stream launches so that every dispatch is
# attributed.  Summaries: gpurun_out/r2_pmc_<workload>_{fetch,write,mfma_util}.csv (copied to profiles/ by the builder,
# together with profiles/pmc_manifest.json naming the kernel-source hash they belong to).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MTD_GRAPH=0 MTD_NO_SIDE_STREAMS=1
O=gpurun_out; mkdir -p $O
for wl in full_step generator; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf $O/pmc_${wl}_$name
    timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_${wl}_$name -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-generator > $O/pmc_${wl}_$name.log 2>&1 || { echo "pmc pass $wl $name failed"; tail -5 $O/pmc_${wl}_$name.log; exit 1; }
    python tools/pmc_summary.py $O/pmc_${wl}_$name $O/r2_pmc_${wl}_$name.csv
    find $O/pmc_${wl}_$name -type f -size +4M -delete
  done
  python tools/pmc_mfma_util.py $O/r2_pmc_${wl}_mfma.csv $O/r2_pmc_${wl}_mfma_util.csv
  head -5 $O/r2_pmc_${wl}_fetch.csv $O/r2_pmc_${wl}_write.csv $O/r2_pmc_${wl}_mfma_util.csv | cut -c1-180
done
python - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
open("gpurun_out/r2_pmc_source_hash.txt", "w").write(bench._kernel_source_hash())
PY

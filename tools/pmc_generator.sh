#!/bin/bash
# PMC passes of the generator workload (separate runs, as MI355X_MICROARCH.md prescribes): HBM-side fetch / write sizes and
# MFMA-pipe busy cycles per kernel.  Eager single-stream launches so that every dispatch is attributed.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MTD_GRAPH=0 MTD_NO_SIDE_STREAMS=1
O=gpurun_out; mkdir -p $O
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf $O/pmc_gen_$name
  timeout -k 10 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_gen_$name -- python3 bench.py --workload generator --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/pmc_gen_$name.log 2>&1
  python tools/pmc_summary.py $O/pmc_gen_$name $O/pmc_generator_$name.csv
  find $O/pmc_gen_$name -type f -size +8M -delete
done
python tools/pmc_mfma_util.py $O/pmc_generator_mfma.csv $O/pmc_generator_mfma_util.csv
head -4 $O/pmc_generator_fetch.csv $O/pmc_generator_write.csv $O/pmc_generator_mfma_util.csv | cut -c1-200

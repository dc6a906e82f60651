#!/bin/bash
# lab: cost attribution of the Res-FFT block tail launch (MTD_TAIL_LAB: bit 0 no DFT MFMAs, bit 1 no spectrum loads; wrong results)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for lab in 0 1 2 3; do
  rm -rf $O/prof_lab$lab
  MTD_TAIL_LAB=$lab MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_lab$lab -o gen -- python3 bench.py --workload generator --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/prof_lab.log 2>&1 || exit 1
  f=$(find $O/prof_lab$lab -name "*.db" | head -1)
  python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps 16 2>/dev/null
  echo "lab=$lab"; grep -E "igemm_c32t" ${f%.db}_kernel_stats.csv | awk -F, "{print substr(\$1,40,40), \$2, \$4}"
done

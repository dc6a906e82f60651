#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_discriminator_gpu.py -m gpu -x -q > $O/pytest_quick.log 2>&1 || { tail -40 $O/pytest_quick.log; exit 1; }
tail -2 $O/pytest_quick.log
rm -rf $O/prof_full
MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $O/prof_full.log 2>&1
find $O/prof_full -name "*.db" | while read f; do python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps 7; done
grep -h "pack_weights\|adamw_kernel" $O/prof_full/*kernel_stats.csv | cut -c1-60,100-200

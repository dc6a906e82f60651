#!/bin/bash
# One GPU-box pass: parity tests, the two bench lines, and the rocprofv3 kernel tables of both workloads.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 560 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 240 python bench.py > $O/bench_full.json 2> $O/bench_full.err
cat $O/bench_full.json | cut -c1-600
timeout -k 10 180 python bench.py --workload generator > $O/bench_gen.json 2> $O/bench_gen.err
cat $O/bench_gen.json | cut -c1-600
rm -rf $O/prof_gen $O/prof_full
MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o gen -- python3 bench.py --workload generator --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/prof_gen.log 2>&1
MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $O/prof_full.log 2>&1
find $O/prof_gen $O/prof_full -name "*.db" | while read f; do python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps $( [[ $f == *gen* ]] && echo 16 || echo 7 ); done
find $O/prof_gen $O/prof_full -name "*.db" -size +30M -delete
ls -la $O/prof_gen/* $O/prof_full/* | head -20

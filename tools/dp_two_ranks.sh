#!/bin/bash
# The data-parallel schedule at world size 2 on the one GPU of a gpurun box (tests/dp_two_ranks_one_gpu.py), then a short
# 2-rank bench rehearsal (MTD_DP_SHARE_GPU=1).  Log -> profiles/r5_dp_two_ranks_one_gpu.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 700 python tests/dp_two_ranks_one_gpu.py > $O/dp2.log 2>&1; rc=$?
grep -v "amdgpu.ids" $O/dp2.log | tail -20
echo "exit code $rc"
[ $rc -eq 0 ] || exit $rc
MTD_DP_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api > $O/dp2_bench.json 2> $O/dp2_bench.err; rc=$?
cut -c1-600 $O/dp2_bench.json; echo "bench exit code $rc"

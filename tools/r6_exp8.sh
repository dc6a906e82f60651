#!/bin/bash
# round 6, experiment 7: single-stream forms of the Winograd block forward / backward in the full step and in the generator leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
MTD_LAB=1 MTD_LAB_LIB=0 MTD_BLOCK_BWD_WINO=3 MTD_BLOCK_FWD_WINO=2 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -x -q -k "oracle or full_batch" > $O/exp8_tests.log 2>&1 || { tail -40 $O/exp8_tests.log; exit 1; }
tail -2 $O/exp8_tests.log
for cfg in "MTD_X=0" "MTD_BLOCK_FWD_WINO=2" "MTD_BLOCK_BWD_WINO=2" "MTD_BLOCK_BWD_WINO=3" "MTD_X=0" "MTD_BLOCK_FWD_WINO=2 MTD_BLOCK_BWD_WINO=3" "MTD_BLOCK_FWD_WINO=2"; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [$cfg] $ms ms"
done | tee $O/exp8_ab_step.txt
for cfg in "MTD_X=0" "MTD_BLOCK_FWD_WINO=2" "MTD_BLOCK_BWD_WINO=3" "MTD_BLOCK_FWD_WINO=2 MTD_BLOCK_BWD_WINO=3" "MTD_X=0"; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 $cfg timeout -k 10 200 python bench.py --workload generator --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "generator [$cfg] $ms ms"
done | tee $O/exp8_ab_gen.txt
# forced data-parallel path on one rank: kernel traces with and without early shipping (where do +0.6 / +0.3 ms go?)
export TMPDIR=/tmp
for tag in ship noship plain; do
  case $tag in ship) E="MTD_FORCE_DP=1";; noship) E="MTD_FORCE_DP=1 MTD_LAB=1 MTD_LAB_LIB=0 MTD_DP_EARLY_SHIP=0";; plain) E="MTD_X=0";; esac
  rm -rf $O/trace_$tag
  env $E timeout -k 10 200 rocprofv3 --kernel-trace -d $O/trace_$tag -o t -- python3 bench.py --steps 6 --warmup 6 $NOX > $O/trace_$tag.log 2>&1 || { echo "trace $tag failed"; tail -5 $O/trace_$tag.log; continue; }
  f=$(find $O/trace_$tag -name "*.db" | head -1)
  echo "== $tag ($E)"; python tools/trace_gaps.py $f --steps 3 --top 12
  find $O/trace_$tag -name "*.db" -size +30M -delete
done | tee $O/exp8_fdp_traces.txt

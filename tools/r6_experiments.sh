#!/bin/bash
# Round 6: every GPU-box experiment of the round as it was run (one function per gpurun call; numbers: profiles/r6_ab_experiments.txt).
#   usage: gpurun -- "bash tools/r6_experiments.sh exp8"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"

# round 6, experiment 1: Res-FFT block conv on the persistent F(2x4) kernel (forward / data gradient) -- parity under the switches, then A/B
exp1() {
MTD_LAB=1 MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -x -q > $O/exp1_tests.log 2>&1 || { tail -30 $O/exp1_tests.log; exit 1; }
tail -2 $O/exp1_tests.log
bash tools/ab_gen.sh "MTD_X=0" "MTD_BLOCK_BWD_WINO=1" 2 | tee $O/exp1_ab.txt
bash tools/ab_gen.sh "MTD_BLOCK_FWD_WINO=1" "MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" 2 | tee -a $O/exp1_ab.txt
}

# round 6, experiment 2: activation-side <G, W> of the spectral-norm correction; Winograd block variants in the full step
exp2() {
timeout -k 10 600 python -m pytest tests/test_discriminator_gpu.py tests/test_step_gpu.py -x -q > $O/exp2_tests.log 2>&1 || { tail -40 $O/exp2_tests.log; exit 1; }
tail -2 $O/exp2_tests.log
bash tools/ab_step.sh "MTD_SN_ACT_DOT=0" "MTD_SN_ACT_DOT=1" 2 | tee $O/exp2_ab.txt
bash tools/ab_step.sh "MTD_BLOCK_BWD_WINO=1" "MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" 2 | tee -a $O/exp2_ab.txt
}

# round 6, experiments 3 + 4: the plain 32 -> 32 layers' data gradients on the persistent F(2x4) kernel's MASKED2 form; fused power iterations
exp3() {
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py -x -q -k "power or spectral or sn_grad or train_forward" > $O/exp4_tests.log 2>&1 || { tail -40 $O/exp4_tests.log; exit 1; }
tail -2 $O/exp4_tests.log
timeout -k 10 400 python -m pytest tests/test_step_gpu.py -x -q > $O/exp4_tests_step.log 2>&1 || { tail -40 $O/exp4_tests_step.log; exit 1; }
tail -2 $O/exp4_tests_step.log
MTD_LAB=1 MTD_WINO_C32_BWD=1 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -x -q -k "oracle or full_batch or replay" > $O/exp3_tests_gen.log 2>&1 || { tail -40 $O/exp3_tests_gen.log; exit 1; }
tail -2 $O/exp3_tests_gen.log
bash tools/ab_step.sh "MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp4_ab.txt
bash tools/ab_gen.sh "MTD_X=0" "MTD_WINO_C32_BWD=1" 2 | tee $O/exp3_ab.txt
bash tools/ab_gen.sh "MTD_WINO_C32_BWD=1 MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" "MTD_X=0" 1 | tee -a $O/exp3_ab.txt
bash tools/ab_step.sh "MTD_X=0" "MTD_WINO_C32_BWD=1" 2 | tee -a $O/exp3_ab.txt
}

# round 6, experiment 4b: fused power iterations (alignment fix), forced-DP one-rank line with the N > 1 diagnostics
exp4() {
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py -x -q -k "power or spectral" > $O/exp4_tests.log 2>&1 || { tail -40 $O/exp4_tests.log; exit 1; }
tail -2 $O/exp4_tests.log
bash tools/ab_step.sh "MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp4_ab.txt
MTD_FORCE_DP=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX > $O/fdp.json 2> $O/fdp.err || { tail -20 $O/fdp.err; exit 1; }
python - <<'PY'
import json
z = json.loads(open("gpurun_out/fdp.json").read().strip().splitlines()[-1])
print({k: z.get(k) for k in ("ms_per_step", "ranks_seen", "allreduce_payload_mb", "allreduce_standalone_ms", "allreduce_alg_gbs", "allreduce_bus_gbs", "launch_mode_per_rank", "ms_per_step_collectives_stubbed", "comm_exposed_ms", "graph_error")})
PY
timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX > $O/plain.json 2> $O/plain.err && python -c "
import json; z=json.loads(open('gpurun_out/plain.json').read().strip().splitlines()[-1]); print('plain', z['ms_per_step'])"
}

# round 6, experiment 4c: fused power iterations with 2 (shipped build) and 1 (lab build, -DMTD_SN_FUSE_GR=1) rows per thread at a time
exp5() {
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py tests/test_generator_gpu.py -x -q -k "power or deferred" > $O/exp5_tests.log 2>&1 || { tail -40 $O/exp5_tests.log; exit 1; }
tail -2 $O/exp5_tests.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=0" "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp5_ab.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee -a $O/exp5_ab.txt
}

# round 6, experiment 5: radix-8 LDS passes of the whole-slice transforms; fused power iterations with 1024-thread workgroups (lab build);
exp6() {
# the price of the Winograd conv's split of K
timeout -k 10 500 python -m pytest tests/test_inference_gpu.py -x -q > $O/exp6_tests.log 2>&1 || { tail -40 $O/exp6_tests.log; exit 1; }
tail -2 $O/exp6_tests.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('inference512 radix-8', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done | tee $O/exp6_inf.txt
MTD_LAB=1 timeout -k 10 300 python tools/splitk_price.py | tee $O/splitk_price.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_X=1" 2 | tee $O/exp6_ab.txt
}

# round 6, experiment 6: radix-8 (shipped build) against radix-4 (lab build, -DMTD_ANY_R8=0) LDS passes on one box; forced-DP with and without early shipping
exp7() {
for i in 1 2 3; do
  for cfg in "MTD_X=0" "MTD_LAB=1"; do
    ms=$(env $cfg timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "inference512 [$cfg: MTD_LAB=1 = radix-4 lab build] $ms ms"
  done
done | tee $O/exp7_inf.txt
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=1 MTD_LAB=1 MTD_LAB_LIB=0 MTD_DP_EARLY_SHIP=0" "MTD_FORCE_DP=0" "MTD_FORCE_DP=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "
import sys,json; z=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', z['ms_per_step'], z.get('ms_per_step_collectives_stubbed'), z.get('comm_exposed_ms'))"
done | tee $O/exp7_fdp.txt
}

# round 6, experiment 7: single-stream forms of the Winograd block forward / backward in the full step and in the generator leg
exp8() {
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
}

# round 6: the GPU suite at the current defaults; four A/B pairs default vs single-stream Winograd block forms; the two-rank rehearsal
exp9() {
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_b.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/gpu_tests_b.log
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_LAB_LIB=0 MTD_BLOCK_FWD_WINO=2 MTD_BLOCK_BWD_WINO=3" 4 | tee $O/exp9_ab.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_LAB_LIB=0 MTD_BLOCK_BWD_WINO=3" 2 | tee -a $O/exp9_ab.txt
bash tools/dp_two_ranks.sh 2>&1 | tail -12 | cut -c1-700
}

# round 6, experiment 8: both halves of a paired pass in one small-map weight-gradient launch (mtd_wgrad_args.half_scale)
exp10() {
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_discriminator_gpu.py -x -q -k "half_scale or sn_grad or weight_gradient_pair or train_forward or spectral" > $O/exp10_tests.log 2>&1 || { tail -40 $O/exp10_tests.log; exit 1; }
tail -2 $O/exp10_tests.log
timeout -k 10 500 python -m pytest tests/test_step_gpu.py tests/test_generator_gpu.py -x -q > $O/exp10_tests_step.log 2>&1 || { tail -40 $O/exp10_tests_step.log; exit 1; }
tail -2 $O/exp10_tests_step.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=0" "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=1" 3 | tee $O/exp10_ab.txt
}

# experiment 11
exp11() {
timeout -k 10 200 python tools/r6_probe_routes.py 2>&1 | grep -v "amdgpu.ids" | tee $O/exp11_routes.txt
MTD_LAB=1 MTD_LAB_LIB=0 MTD_BLOCK_FWD_WINO=0 MTD_BLOCK_BWD_WINO=0 timeout -k 10 200 python tools/r6_probe_routes.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/exp11_routes.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=0" "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=1" 3 | tee $O/exp10_ab.txt
}

"$@"

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

# round 6, experiment 9: the two decoders' mirror convs in pairs (kernels.conv_pair / wino_conv_multi_kernel)
exp13() {
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_pair or winograd_conv_vs_torch" > $O/exp13_tests.log 2>&1 || { tail -40 $O/exp13_tests.log; exit 1; }
tail -2 $O/exp13_tests.log
timeout -k 10 600 python -m pytest tests/test_discriminator_gpu.py tests/test_step_gpu.py -x -q > $O/exp13_tests_step.log 2>&1 || { tail -40 $O/exp13_tests_step.log; exit 1; }
tail -2 $O/exp13_tests_step.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_PAIR_DECODERS=0" "MTD_LAB_LIB=0 MTD_PAIR_DECODERS=1" 3 | tee $O/exp13_ab.txt
}

# round 6, experiment 10: the adversarial and the first consistency pass advanced together (discriminator_path.disc_backward_lockstep)
exp14() {
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_discriminator_gpu.py -x -q > $O/exp14_tests.log 2>&1 || { tail -40 $O/exp14_tests.log; exit 1; }
tail -2 $O/exp14_tests.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=0" "MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=1" 3 | tee $O/exp14_ab.txt
}

# round 6, experiment 11: pair launches only up to a map size (MTD_PAIR_MAX_PIXELS)
exp15() {
for rep in 1 2; do
for px in 0 4096 16384 65536; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 MTD_PAIR_MAX_PIXELS=$px timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_PAIR_MAX_PIXELS=$px] $ms ms"
done
done | tee $O/exp15_ab.txt
}

# round 6, final checks: the GPU suite + smoke at the final tree, the forced data-parallel line beside the plain one, the two-rank rehearsal
exp16() {
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=0" "MTD_FORCE_DP=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null > $O/fdp_$cfg.json
  python -c "
import sys,json; z=json.loads(open('$O/fdp_$cfg.json').read().strip().splitlines()[-1]); print('[$cfg]', z['ms_per_step'], z.get('ms_per_step_collectives_stubbed'), z.get('comm_exposed_ms'), z.get('graph_error'))"
done | tee $O/exp16_fdp.txt
bash tools/dp_two_ranks.sh 2>&1 | grep "dp2\|exit code" | cut -c1-400
}

# round 6, experiment 12b: the pair launches' split of K planned for 2 (shipped), 3, 4 problems' worth of grid (lab library)
exp17() {
for rep in 1 2; do
for v in 2 3 4 1; do
  ms=$(env MTD_LAB=1 MTD_WINO_PAIR_SPLIT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_WINO_PAIR_SPLIT=$v] $ms ms"
done
done | tee $O/exp17b_ab.txt
}

# round 6, experiment 13b: passes advanced together -- none (0), two (2), three (1) -- on one box
exp18() {
for rep in 1 2 3; do
for v in 0 2 1; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_LOCKSTEP_PASSES=$v] $ms ms"
done
done | tee $O/exp18b_ab.txt
}

# round 6: the GPU suite + smoke at the final tree, then the measurement pipeline
exp19() {
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; rc=$?; echo "gpu suite exit $rc"; tail -3 $O/gpu_tests_final.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/measure.sh r6 > $O/measure_final.log 2>&1; tail -3 $O/measure_final.log | cut -c1-300
}

# round 6, experiment 14 (lab library): three passes advanced together with the groups planned as pairs (MTD_WINO_PAIR_SPLIT=2) against the shipped two
exp20() {
for rep in 1 2 3; do
for cfg in "MTD_LAB_LIB=0" "MTD_LOCKSTEP_PASSES=3 MTD_WINO_PAIR_SPLIT=2" "MTD_LOCKSTEP_PASSES=3"; do
  ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [$cfg] $ms ms"
done
done | tee $O/exp20_ab.txt
}

# round 6, experiment 15: which early-shipping point costs the forced data-parallel path its 0.34 ms at one rank?
exp21() {
for rep in 1 2; do
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=1 MTD_DP_SHIP_STAGES=heads" "MTD_FORCE_DP=1 MTD_DP_SHIP_STAGES=trunk_low" "MTD_FORCE_DP=1 MTD_DP_EARLY_SHIP=0"; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [$cfg] $ms ms"
done
done | tee $O/exp21_ab.txt
}

# round 6, experiment 16: the thin layers' two slab-sum stages in one launch (lab library: MTD_WGRAD_SCALAR2=0 is the two-launch form)
# bit-identity of the one-launch form against the two-launch form on the thin layers' shapes (two processes: the lab switch is read once)
exp22() {
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "direct or thin or golden or c1 or n1 or wgrad" > $O/exp22_tests.log 2>&1 || { tail -30 $O/exp22_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp22_tests.log
python - <<'PY'
import subprocess, sys, os
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from mtd_gan_amd import kernels as K
outs = []
for (B, Ci, Co, H, k) in ((64, 1, 64, 64, 3), (64, 128, 1, 64, 3), (64, 1, 1, 64, 3), (64, 512, 1, 1, 1), (32, 1, 32, 64, 3)):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, H, Ci, generator=g).cuda(); gy = torch.randn(B, H, H, Co, generator=g).cuda()
    dw = torch.empty(Co, Ci, k, k, device="cuda"); db = torch.empty(Co, device="cuda")
    K.wgrad(gy, x, K.geom_fwd(B, H, H, k, 1, (k - 1) // 2), Co, Ci, dw, Ci * k * k, k * k, db=db)
    torch.cuda.synchronize()
    outs.append(dw.cpu()); outs.append(db.cpu())
torch.save(outs, sys.argv[1])
'''
for tag, env in (("one", {}), ("two", {"MTD_WGRAD_SCALAR2": "0"})):
    subprocess.run([sys.executable, "-c", code, f"/tmp/scalar2_{tag}.pt"], env=dict(os.environ, MTD_LAB="1", **env), check=True)
import torch
a, b = torch.load("/tmp/scalar2_one.pt"), torch.load("/tmp/scalar2_two.pt")
print("thin-layer weight gradients, one launch == two launches bit for bit:", all(torch.equal(x, y) for x, y in zip(a, b)), len(a), "tensors")
PY
bash tools/ab_step.sh "MTD_WGRAD_SCALAR2=0" "MTD_WGRAD_SCALAR2=1" 3 | tee $O/exp22_ab.txt
}

# round 6, experiment 17: the restoration pass's trunk advanced together with the second consistency pass's (MTD_LOCKSTEP_PASSES=2)
exp23() {
timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -k "paired_launches" > $O/exp23_tests.log 2>&1 || { tail -30 $O/exp23_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp23_tests.log
bash tools/ab_step.sh "MTD_LOCKSTEP_PASSES=1" "MTD_LOCKSTEP_PASSES=2" 3 | tee $O/exp23_ab.txt
}

# round 6, experiment 18: plan of the 4x4 stride-2 weight gradients on small maps after the merge of the halves (MTD_WGRAD_T16_PLAN=0: round 5's thresholds)
exp24() {
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "wgrad or golden or sn_grad" > $O/exp24_tests.log 2>&1 || { tail -30 $O/exp24_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp24_tests.log
bash tools/ab_step.sh "MTD_WGRAD_T16_PLAN=0" "MTD_WGRAD_T16_PLAN=1" 3 | tee $O/exp24_ab.txt
}

# round 6: the concurrent step's timeline (recorded list, side streams): wall / busy union / idle / sum of kernel durations
exp25() {
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_conc -o conc -- python3 bench.py --steps 6 --warmup 4 $NOX > $O/prof_conc.log 2>&1 || { tail -5 $O/prof_conc.log; exit 1; }
f=$(find $O/prof_conc -name "*.db" | head -1)
python tools/trace_gaps.py $f --steps 4 --top 12 | tee $O/r6_concurrent_timeline.txt
find $O/prof_conc -name "*.db" -size +30M -delete
}

# round 6, end of round: the forced data-parallel line beside the plain one and the two-rank rehearsal at the final tree
exp26() {
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=0" "MTD_FORCE_DP=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null > $O/fdp_$cfg.json
  python -c "
import sys,json; z=json.loads(open('$O/fdp_$cfg.json').read().strip().splitlines()[-1]); print('[$cfg]', z['ms_per_step'], z.get('ms_per_step_collectives_stubbed'), z.get('comm_exposed_ms'), z.get('graph_error'))"
done | tee $O/exp26_fdp.txt
bash tools/dp_two_ranks.sh > $O/exp26_dp2.txt 2>&1; grep "dp2\|exit code" $O/exp26_dp2.txt | cut -c1-400
}

# round 6, experiment 20: whole-slice column kernel, the two real columns kw = 0 and kw = S/2 of an image through one packed transform (lab library: MTD_ANY_PACK=0 = units of their own)
exp27() {
timeout -k 10 500 python -m pytest tests/test_inference_gpu.py -x -q > $O/exp27_tests.log 2>&1 || { tail -40 $O/exp27_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp27_tests.log
for i in 1 2 3; do
  for cfg in "MTD_ANY_PACK=0" "MTD_ANY_PACK=1"; do
    ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "inference512 [$cfg] $ms ms"
  done
done | tee $O/exp27_inf.txt
}

# round 6, experiment 21: whole-slice spectral kernels walking their lines from the last image to the first (lab library MTD_ANY_REV: bit 0 rows, 1 columns, 2 rows back)
exp28() {
MTD_LAB=1 MTD_ANY_REV=7 timeout -k 10 500 python -m pytest tests/test_inference_gpu.py -x -q > $O/exp28_tests.log 2>&1 || { tail -40 $O/exp28_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp28_tests.log
for i in 1 2; do
  for v in 0 2 6 4 7 5 3 1; do
    ms=$(env MTD_LAB=1 MTD_ANY_REV=$v timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "inference512 [MTD_ANY_REV=$v] $ms ms"
  done
done | tee $O/exp28_inf.txt
}

"$@"

#!/bin/bash
# the N > 1 code path on the one GPU of a box: every collective forced in a one-rank group, and the launcher's environment
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
MTD_FORCE_DP=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/dp_force.json 2> $O/dp_force.err || { tail -20 $O/dp_force.err; exit 1; }
cut -c1-330 $O/dp_force.json
MTD_FORCE_DP=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/dp_torchrun.json 2> $O/dp_torchrun.err || { tail -20 $O/dp_torchrun.err; exit 1; }
cut -c1-330 $O/dp_torchrun.json
MTD_FORCE_DP=1 timeout -k 10 200 python bench.py --workload generator --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/dp_force_gen.json 2> $O/dp_force_gen.err || { tail -20 $O/dp_force_gen.err; exit 1; }
cut -c1-330 $O/dp_force_gen.json

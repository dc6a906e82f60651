#!/bin/bash
# round 6, experiment 16: the thin layers' two slab-sum stages in one launch (lab library: MTD_WGRAD_SCALAR2=0 is the two-launch form)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "direct or thin or golden or c1 or n1 or wgrad" > $O/exp22_tests.log 2>&1 || { tail -30 $O/exp22_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp22_tests.log
python - <<'PY'
# bit-identity of the one-launch form against the two-launch form on the thin layers' shapes (two processes: the lab switch is read once)
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

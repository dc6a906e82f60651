#!/usr/bin/env python3
"""Times the weight gradients of the degenerate channel counts (wgrad_wide_kernel: min(N, C) == 1) at the shapes of the
training step, back to back on one stream: us per launch (main kernel + its slab sum) and algorithmic TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import kernels as K

dev = torch.device("cuda", 0)
CASES = [("D conv11 1->64 (half)", 32, 1, 64), ("D head 128->1 (half)", 32, 128, 1), ("G conv 1->32", 32, 1, 32), ("G conv 32->1", 32, 32, 1)]
REPS = int(os.environ.get("REPS", "50"))
for name, B, Ci, Co in CASES:
    p = torch.randn(B, 64, 64, Co, device=dev)
    q = torch.randn(B, 64, 64, Ci, device=dev)
    dw = torch.empty(Co, Ci, 3, 3, device=dev)
    db = torch.empty(Co, device=dev)
    geom = K.geom_fwd(B, 64, 64, 3, 1, 1)
    run = lambda: K.wgrad(p, q, geom, Co, Ci, dw, Ci * 9, 9, db=db)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / REPS
    nbytes = 4.0 * (p.numel() + q.numel())
    print(f"{name:24s} {us:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 1e6:6.2f} TB/s")

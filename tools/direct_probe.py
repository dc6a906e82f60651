#!/usr/bin/env python3
"""Times the vector-ALU / tap-plane convolutions of the degenerate channel counts (C == 1 or N == 1: mtd_conv_direct) at the
shapes of the training step, back to back on one stream, and prints us per launch and the algorithmic TB/s
(input + output + weights once).  B is the batch of the launch (64 = a paired discriminator pass at batch 32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import kernels as K

dev = torch.device("cuda", 0)
CASES = [  # name, B, Cin, Cout, k
    ("G head conv 1->32", 32, 1, 32, 3), ("D conv11 1->64", 64, 1, 64, 3), ("D conv11 1->64 (B=32)", 32, 1, 64, 3),
    ("G tail conv 32->1", 32, 32, 1, 3), ("D dgrad conv11 64->1", 64, 64, 1, 3), ("D head 128->1", 64, 128, 1, 3),
    ("D head 64->1 1x1", 64, 64, 1, 1),
]
REPS = int(os.environ.get("REPS", "50"))
for name, B, Ci, Co, k in CASES:
    x = torch.randn(B, 64, 64, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    out = torch.empty(B, 64, 64, Co, device=dev)
    geom = K.geom_fwd(B, 64, 64, k, 1, (k - 1) // 2)
    run = lambda: K.conv(x, w, geom, Co, Ci, Ci * k * k, k * k, out, act=1)
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
    nbytes = 4.0 * (x.numel() + out.numel() + w.numel())
    print(f"{name:28s} B={B:3d} {us:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 1e6:6.2f} TB/s")

#!/usr/bin/env python3
"""F(2x4, 3x3) (the plan on maps >= 8 pixels wide) against F(2x2, 3x3) with its two-workgroups-per-CU form on the short-K / large-map
layers of the discriminator, after the round-5 load spreading: us per launch, hot.  (The plan stands: profiles/r5_load_spreading.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtd_gan_amd
from mtd_gan_amd import kernels as K

def timed(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters

dev = torch.device("cuda", 0)
for (B, H, Ci, Co) in ((64, 64, 64, 64), (32, 64, 64, 64), (64, 32, 64, 256), (64, 32, 128, 128), (64, 32, 256, 64), (64, 32, 128, 64)):
    x = torch.randn(B, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    out = torch.empty(B, H, H, Co, device=dev)
    g = K.geom_fwd(B, H, H, 3, 1, 1)
    res = []
    for mw in (8, 0):
        old = K.winograd_f4_min_w(mw)
        K.weights_changed(None)
        t = timed(lambda: K.conv(x, w, g, Co, Ci, Ci * 9, 9, out, bias=b, act=K.ACT_LRELU))
        res.append(t)
        K.winograd_f4_min_w(old)
    print(f"M={B*H*H:7d} N={Co:4d} C={Ci:4d}  F(2x4) {res[0]:7.1f} us   F(2x2) {res[1]:7.1f} us")

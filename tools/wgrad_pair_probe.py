#!/usr/bin/env python3
"""Lab: mtd_conv_wgrad_pair against two mtd_conv_wgrad launches on the discriminator's paired layers (B = 64 images = two halves
of 32; hot inputs, back-to-back launches, event-timed).  usage: tools/wgrad_pair_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import kernels as K

CH = [64, 128, 256, 512, 512, 512]
SHAPES = []
for l in range(1, 7):
    h = 64 >> (l - 1)
    co = CH[l - 1]
    ci = 1 if l == 1 else CH[l - 2]
    if ci >= 32:
        SHAPES.append((f"conv{l}1", ci, co, h, 3))
    SHAPES.append((f"conv{l}2", co, co, h, 3))
    SHAPES.append((f"down{l}", co, co, h // 2, 4))
SHAPES.append(("bconv", 512, 512, 1, 1))
B, Bh = 64, 32
for name, Ci, Co, H, k in SHAPES:
    stride, pad = (2, 1) if k == 4 else (1, (k - 1) // 2)
    x = torch.randn(B, H * stride, H * stride, Ci, device="cuda")
    gy = torch.randn(B, H, H, Co, device="cuda")
    geom = K.geom_fwd(B, H * stride, H * stride, k, stride, pad)
    gh = K.geom_fwd(Bh, H * stride, H * stride, k, stride, pad)
    kk = k * k
    d1, d2 = torch.empty(Co, Ci, k, k, device="cuda"), torch.empty(Co, Ci, k, k, device="cuda")
    db = torch.zeros(Co, device="cuda")

    def pair():
        K.wgrad_pair(gy, x, geom, Bh, Co, Ci, d1, d2, Ci * kk, kk, db=db, accumulate_bias=True)

    def two():
        K.wgrad(gy[:Bh], x[:Bh], gh, Co, Ci, d1, Ci * kk, kk, db=db, accumulate=False, accumulate_bias=True)
        K.wgrad(gy[Bh:], x[Bh:], gh, Co, Ci, d2, Ci * kk, kk, db=db, accumulate=False, accumulate_bias=True)
    res = {}
    for nme, fn in (("pair", pair), ("two", two)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[nme] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:8s} C{Ci:4d} N{Co:4d} {H:3d}x{H:<3d} k{k}  pair {res['pair']:7.1f} us   two launches {res['two']:7.1f} us", flush=True)

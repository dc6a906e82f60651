#!/usr/bin/env python3
"""Weight gradient of the generator's 32 -> 32 channel 3x3 layers at the training size (32 / 64 patches of 64 x 64): the Winograd
F(2x2, 3x3) kernel on one 32 x 32 block (plan 19, csrc/conv_wgrad_wino32.h) against the row-window kernel (plans 7 / 8), forward and
transposed tap order; us per call including the slab sum, hot, and the error of both against float64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    for B in (32, 64):
        x = torch.randn(B, 64, 64, 32, device=dev)
        gy = torch.randn(B, 64, 64, 32, device=dev)
        wc = torch.zeros(32, 32, 3, 3, dtype=torch.double, device=dev, requires_grad=True)
        (F.conv2d(x.permute(0, 3, 1, 2).double(), wc, None, padding=1) * gy.permute(0, 3, 1, 2).double()).sum().backward()
        for name, geom, sn, sc, row_cfg in (("forward", K.geom_fwd(B, 64, 64, 3, 1, 1), 288, 9, 7), ("transposed", K.geom_dgrad_s1(B, 64, 64, 3, 1), 9, 288, 8)):
            res = {}
            for cfg in (row_cfg, 19):
                L.mtd_conv_wgrad_override(cfg, -1)
                dw = torch.empty(32, 32, 3, 3, device=dev)
                db = torch.empty(32, device=dev)
                fn = lambda: K.wgrad(gy, x, geom, 32, 32, dw, sn, sc, db=db)
                res[cfg] = (timed(fn), dw.clone(), db.clone())
            L.mtd_conv_wgrad_override(-1, -1)
            d = (res[row_cfg][1] - res[19][1]).abs().max().item() / res[row_cfg][1].abs().max().item()
            dbd = (res[row_cfg][2] - res[19][2]).abs().max().item() / res[row_cfg][2].abs().max().item()
            print(f"B={B} {name:10s} row-window {res[row_cfg][0]:7.1f} us   Winograd 32x32 {res[19][0]:7.1f} us   difference {d:.1e} (bias {dbd:.1e})")
            if name == "forward":
                for cfg in (row_cfg, 19):
                    print(f"   plan {cfg:2d} error vs float64: {((res[cfg][1].double() - wc.grad).abs().max() / wc.grad.abs().max()).item():.2e}")


if __name__ == "__main__":
    main()

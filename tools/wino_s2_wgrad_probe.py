#!/usr/bin/env python3
"""F(3x3, 2x2) weight gradient of the 4x4 / stride-2 layers (csrc/conv_wgrad_wino_s2.h, plan 18) against the halo-window kernel
(plan 15) on the discriminator's down1..3 shapes: single launches (64 / 32 images) and the pair form (2 x 32 images), us per call
including the slab sum, hot."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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
    print("shape                                   form     halo-window us   wino us   ratio")
    for B in (64, 32):
        for Ci, H in ((64, 64), (128, 32), (256, 16)):
            Co = Ci
            OH = H // 2
            x = torch.randn(B, H, H, Ci, device=dev)
            gy = torch.randn(B, OH, OH, Co, device=dev)
            geom = K.geom_fwd(B, H, H, 4, 2, 1)
            dw, dw2 = torch.empty(Co, Ci, 4, 4, device=dev), torch.empty(Co, Ci, 4, 4, device=dev)
            db = torch.empty(Co, device=dev)
            res = {}
            for form in ("single", "pair"):
                if form == "pair" and B != 64:
                    continue
                for cfg in (15, 18):
                    L.mtd_conv_wgrad_override(cfg, -1)
                    if form == "single":
                        fn = lambda: K.wgrad(gy, x, geom, Co, Ci, dw, Ci * 16, 16, db=db)
                    else:
                        fn = lambda: K.wgrad_pair(gy, x, geom, B // 2, Co, Ci, dw, dw2, Ci * 16, 16, db=db, accumulate_bias=False)
                    res[cfg] = timed(fn)
                    L.mtd_conv_wgrad_override(-1, -1)
                print(f"B={B:3d} {Ci:4d}->{Co:4d} out {OH:2d}x{OH:<2d} M={B * OH * OH:6d}  {form:7s} {res[15]:12.1f} {res[18]:12.1f} {res[15] / res[18]:7.2f}")


if __name__ == "__main__":
    main()

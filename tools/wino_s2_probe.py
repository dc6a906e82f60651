#!/usr/bin/env python3
"""F(3x3, 2x2) kernel of the 4x4 / stride-2 layers (csrc/conv_wino_s2.h) against the implicit GEMM on the discriminator's
down1..3 shapes at the training batch (64 images per D pass, 32 per G pass): forward conv and four-parity data gradient,
us per launch (hot, repeated), and the error of both against float64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import kernels as K


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


def err(a, b):
    return ((a.double() - b).abs().max() / b.abs().max()).item()


def main():
    dev = torch.device("cuda", 0)
    print("shape                               form      igemm us   wino us   ratio   err igemm   err wino")
    for B in (64, 32):
        for Ci, Co, H in ((64, 64, 64), (128, 128, 32), (256, 256, 16), (512, 512, 8)):
            torch.manual_seed(1)
            x = torch.randn(B, H, H, Ci, device=dev)
            w = torch.randn(Co, Ci, 4, 4, device=dev) * 0.05
            b = torch.randn(Co, device=dev)
            OH = H // 2
            geom = K.geom_fwd(B, H, H, 4, 2, 1)
            ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1), 0.2).permute(0, 2, 3, 1)
            out = torch.empty(B, OH, OH, Co, device=dev)
            res = {}
            for form in ("igemm", "wino"):
                K.WINO_S2 = 2 if form == "wino" else 0
                K.WINO_S2_MIN_HW = 2
                fn = lambda: K.conv(x, w, geom, Co, Ci, Ci * 16, 16, out, bias=b, act=K.ACT_LRELU)
                t = timed(fn)
                res[form] = (t, err(out, ref))
            print(f"B={B:3d} {Ci:4d}->{Co:4d} {H:3d}x{H:<3d} M={B * OH * OH:6d}  fwd     {res['igemm'][0]:9.1f} {res['wino'][0]:9.1f} {res['igemm'][0] / res['wino'][0]:7.2f}"
                  f"   {res['igemm'][1]:.2e}   {res['wino'][1]:.2e}")
            cot = torch.randn(B, OH, OH, Co, device=dev)
            # (as in the step: a residual operand and the previous layer's activation mask in the epilogue)
            add = torch.randn(B, H, H, Ci, device=dev)
            msk = torch.randn(B, H, H, Ci, device=dev)
            want = (F.conv_transpose2d(cot.permute(0, 3, 1, 2).double(), w.double(), None, stride=2, padding=1).permute(0, 2, 3, 1) + add.double()) * torch.where(msk > 0, 1.0, 0.2).double()
            dx = torch.empty(B, H, H, Ci, device=dev)
            kw = dict(add1=add, mask=msk, mask_slope=0.2)
            calls = [((cot, w, K.geom_dgrad_s2(B, H, H, py, px), Ci, Co, 16, Ci * 16, dx), dict(kw)) for py in range(2) for px in range(2)]
            for form in ("igemm", "wino"):
                K.WINO_S2 = 2 if form == "wino" else 0
                fn = lambda: K.conv_multi(calls)
                t = timed(fn)
                res[form] = (t, err(dx, want))
            print(f"B={B:3d} {Co:4d}->{Ci:4d} {H:3d}x{H:<3d} M={B * OH * OH:6d}x4 dgrad  {res['igemm'][0]:9.1f} {res['wino'][0]:9.1f} {res['igemm'][0] / res['wino'][0]:7.2f}"
                  f"   {res['igemm'][1]:.2e}   {res['wino'][1]:.2e}")


if __name__ == "__main__":
    main()

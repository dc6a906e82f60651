#!/usr/bin/env python3
"""What a Winograd F(2x2, 3x3) form of the heavy 3x3 stride-1 layers would cost with the kernels that exist: the 16
transform-domain products of a layer are a 1x1 convolution over 16 * M / 4 = 4 M pixels (timed here with the implicit-GEMM
kernel, one weight matrix standing in for the 16), the three transforms are HBM passes (5 M C + 9 M N floats unfused).
Prints the direct 3x3 launch beside it.  Lab tool for DESIGN 9.7; nothing in the product path uses it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import kernels as K

dev = torch.device("cuda", 0)


def timed(run, reps=30):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for (B, H, Ci, Co) in [(64, 16, 256, 256), (64, 32, 128, 128), (64, 64, 64, 64), (64, 8, 512, 512), (64, 16, 128, 512)]:
    M = B * H * H
    x = torch.randn(B, H, H, Ci, device=dev)
    w3 = torch.randn(Co, Ci, 3, 3, device=dev) * 0.02
    y = torch.empty(B, H, H, Co, device=dev)
    g3 = K.geom_fwd(B, H, H, 3, 1, 1)
    t3 = timed(lambda: K.conv(x, w3, g3, Co, Ci, Ci * 9, 9, y))
    # transform domain: 16 products over M / 4 tiles each = one 1x1 conv over 4 M "pixels"
    xv = torch.randn(4 * B, H, H, Ci, device=dev)
    w1 = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
    yv = torch.empty(4 * B, H, H, Co, device=dev)
    g1 = K.geom_fwd(4 * B, H, H, 1, 1, 0)
    t1 = timed(lambda: K.conv(xv, w1, g1, Co, Ci, Ci, 1, yv))
    traffic = 4.0 * (5 * M * Ci + 9 * M * Co)
    tt = traffic / 4e6          # us at 4 TB/s
    print(f"M={M:6d} C={Ci:4d} N={Co:4d}: direct 3x3 {t3:7.1f} us ({2 * 9 * M * Ci * Co / t3 / 1e6:6.1f} TF/s) | transform-domain GEMM {t1:7.1f} us "
          f"({2 * 4 * M * Ci * Co / t1 / 1e6:6.1f} TF/s) + unfused transforms {traffic / 1e6:6.1f} MB = {tt:5.1f} us at 4 TB/s -> {t1 + tt:7.1f} us")

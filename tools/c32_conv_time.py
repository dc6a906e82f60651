"""Time one 32 -> 32 channel 3x3 layer of the generator on a whole-slice batch (8 x 512 x 512, the inference512 bench's size) or
on the training batch (32 x 64 x 64):  python tools/c32_conv_time.py [B H W] .  Forms: the persistent F(2x4) kernel
(csrc/conv_wino_c32.h), the general Winograd kernel's 32-channel workgroups (MTD_WINO_C32_KERNEL=0), the implicit GEMM
(MTD_WINOGRAD=0) -- one process per form (the switches are read once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402,F401
from mtd_gan_amd import kernels as K  # noqa: E402


def main():
    B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 512, 512)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, 32, generator=g).cuda()
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).cuda()
    bias, add1 = torch.zeros(32).cuda(), torch.randn(B, H, W, 32, generator=g).cuda()
    geom = K.geom_fwd(B, H, W, 3, 1, 1)
    out = torch.empty_like(x)
    for name, kw in (("relu(conv + b)", dict(bias=bias, act=K.ACT_RELU)), ("relu(conv + b) + x", dict(bias=bias, add1=add1, act=K.ACT_RELU_ADD)),
                     ("relu(conv + b + skip)", dict(bias=bias, add1=add1, act=K.ACT_RELU))):
        if kw["act"] == K.ACT_RELU_ADD and not K.conv_relu_add_ok(x, w, geom, 32, 32, 288, 9, out, bias=bias, add1=add1):
            continue
        for _ in range(3):
            K.conv(x, w, geom, 32, 32, 288, 9, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            K.conv(x, w, geom, 32, 32, 288, 9, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        px = B * H * W
        nbytes = px * 128 * (3 if "add1" in kw else 2)
        print(f"{B}x{H}x{W} {name:22s} {us:8.1f} us   {2.0 * px * 32 * 32 * 9 / us * 1e-6:6.1f} TFLOP/s direct-equivalent   {nbytes / us * 1e-3:6.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()

"""Micro-benchmark of single kernels (for rocprofv3 --pmc runs): the generator's 3x3 32->32 conv at B=32,
its data gradient and its weight gradient, N repetitions each, HIP-event timing printed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mtd_gan_amd import kernels as K  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, H, W, C = 32, 64, 64, 32
dev = torch.device("cuda")
x = torch.randn(B, H, W, C, device=dev)
g = torch.randn(B, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.06
b = torch.randn(C, device=dev) * 0.1
out = torch.empty(B, H, W, C, device=dev)
dw = torch.empty_like(w)
db = torch.empty_like(b)
gf, gd = K.geom_fwd(B, H, W, 3, 1, 1), K.geom_dgrad_s1(B, H, W, 3, 1)


def timeit(name, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} {us:9.1f} us   {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


F = 2.0 * B * H * W * C * C * 9
if which in ("all", "fwd"):
    timeit("conv3x3 fwd (+bias,relu)", lambda: K.conv(x, w, gf, C, C, C * 9, 9, out, bias=b, act=K.ACT_RELU), F)
if which in ("all", "dgrad"):
    timeit("conv3x3 dgrad (+add)", lambda: K.conv(g, w, gd, C, C, 9, C * 9, out, add1=x), F)
if which in ("all", "wgrad"):
    timeit("conv3x3 wgrad", lambda: K.wgrad(g, x, gf, C, C, dw, C * 9, 9, db=db), F)
if which in ("all", "rows"):
    R = K.rfft_rows(x, 0)
    timeit("rfft_rows", lambda: K.rfft_rows(x, 0), 1)
    timeit("irfft_rows(+2 adds,mask)", lambda: K.irfft_rows(R, out, add1=x, add2=g, mask=x), 1)
    w2 = torch.randn(64, 64, device=dev) * 0.1
    b2 = torch.randn(64, device=dev) * 0.1
    w2t = K.transpose64(w2)
    T, S, Z = K.spec_mix_fwd(R, w2t, b2, True)
    timeit("spec_mix_fwd", lambda: K.spec_mix_fwd(R, w2t, b2, True), 2.0 * B * 2112 * 64 * 64)
    dw2 = torch.empty(64, 64, device=dev)
    db2 = torch.empty(64, device=dev)
    timeit("spec_mix_bwd(+reduce)", lambda: K.spec_mix_bwd(R, w2, S, Z, dw2, db2), 2.0 * 2 * B * 2112 * 64 * 64)
if which in ("all", "taps"):
    # per-chunk cost vs fixed cost of the implicit-GEMM kernel: same tensors, 1 / 9 / 16(4x4,pad 1... use 3x3 twice) taps
    w1 = torch.randn(C, C, 1, 1, device=dev) * 0.1
    timeit("conv1x1 fwd (1 chunk)", lambda: K.conv(x, w1, K.geom_fwd(B, H, W, 1, 1, 0), C, C, C, 1, out), 2.0 * B * H * W * C * C)
    timeit("conv3x3 fwd plain", lambda: K.conv(x, w, gf, C, C, C * 9, 9, out), F)
    x64 = torch.randn(B, H, W, 64, device=dev)
    w64 = torch.randn(32, 64, 3, 3, device=dev) * 0.05
    timeit("conv3x3 64->32 (18 chunks)", lambda: K.conv(x64, w64, gf, 32, 64, 64 * 9, 9, out), 2 * F)
    o64 = torch.empty(B, H, W, 64, device=dev)
    w6464 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    timeit("conv3x3 64->64 <2,2,4,1>", lambda: K.conv(x64, w6464, gf, 64, 64, 64 * 9, 9, o64), 4 * F)
    x128 = torch.randn(B, 32, 32, 128, device=dev)
    o128 = torch.empty(B, 32, 32, 128, device=dev)
    w128 = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    g32 = K.geom_fwd(B, 32, 32, 3, 1, 1)
    timeit("conv3x3 128->128 @32 <2,2,2,2>", lambda: K.conv(x128, w128, g32, 128, 128, 128 * 9, 9, o128), 2.0 * B * 32 * 32 * 128 * 128 * 9)
    x512 = torch.randn(B, 8, 8, 512, device=dev)
    o512 = torch.empty(B, 8, 8, 512, device=dev)
    w512 = torch.randn(512, 512, 3, 3, device=dev) * 0.02
    g8 = K.geom_fwd(B, 8, 8, 3, 1, 1)
    timeit("conv3x3 512->512 @8 (splitK)", lambda: K.conv(x512, w512, g8, 512, 512, 512 * 9, 9, o512), 2.0 * B * 8 * 8 * 512 * 512 * 9)
    timeit("empty-ish: act_grad 16MiB", lambda: K.act_grad(g, x, 0.0, out), 1)
if which in ("all", "locality"):
    from mtd_gan_amd._lib import Geom
    def geom(dy, dx):
        return Geom(B, H, W, H, W, 1, 1, 0, 0, dy, dx, 3, 3, 3, 0, 0, 1, 1, H, W, 1, 1, 0, 0)
    timeit("9 taps, all same pixel", lambda: K.conv(x, w, geom(0, 0), C, C, C * 9, 9, out), F)
    timeit("9 taps, dx only", lambda: K.conv(x, w, geom(0, 1), C, C, C * 9, 9, out), F)
    timeit("9 taps, dy only", lambda: K.conv(x, w, geom(1, 0), C, C, C * 9, 9, out), F)
    timeit("9 taps, 3x3 (regular)", lambda: K.conv(x, w, gf, C, C, C * 9, 9, out), F)
if which in ("all", "zeros"):
    xz, wz = torch.zeros_like(x), torch.zeros_like(w)
    timeit("3x3 zeros data+weights", lambda: K.conv(xz, wz, gf, C, C, C * 9, 9, out), F)
    timeit("3x3 random", lambda: K.conv(x, w, gf, C, C, C * 9, 9, out), F)
    timeit("3x3 zeros data+weights", lambda: K.conv(xz, wz, gf, C, C, C * 9, 9, out), F)
    timeit("3x3 random", lambda: K.conv(x, w, gf, C, C, C * 9, 9, out), F)

"""Round 6 probe: the deep-level launches of the discriminator (1x1 / 2x2 / 4x4 maps, 512 channels) under different splits of K:
us per conv (kernel + split-K epilogue), hot operands.   python tools/small_map_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtd_gan_amd import kernels as K


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


scale = torch.tensor([0.7], device="cuda")
print("1x1 conv 512 -> 512 on 1x1 maps (bconv1 / bconv2 / c_fc): igemm_kernel<1,1,4,1> (cfg 1) and others, forced split of K")
for B in (64, 32):
    x = torch.randn(B, 1, 1, 512, device="cuda")
    w = torch.randn(512, 512, 1, 1, device="cuda") * 0.04
    out = torch.empty(B, 1, 1, 512, device="cuda")
    bias = torch.randn(512, device="cuda")
    geom = K.geom_fwd(B, 1, 1, 1, 1, 0)
    row = f"M={B:4d}: plan {timed(lambda: K.conv(x, w, geom, 512, 512, 512, 1, out, bias=bias, act=K.ACT_LRELU, scale=scale)):6.1f} us"
    for cfg in (1, 5):
        for sk in (1, 2, 4, 8, 16):
            K.igemm_override(cfg, sk)
            try:
                row += f" | cfg{cfg} S={sk}: {timed(lambda: K.conv(x, w, geom, 512, 512, 512, 1, out, bias=bias, act=K.ACT_LRELU, scale=scale)):5.1f}"
            finally:
                K.igemm_override(-1, -1)
    print(row, flush=True)

# ---- weight gradient of the 4x4 stride-2 layers on small maps under every plan the library can be forced to (mtd_conv_wgrad_override): us per call
# (kernel + slab sums) and the result against the plan's; both halves of a paired pass in one launch = 64 images (down4: 8x8 -> 4x4, down5: 4x4 -> 2x2)
from mtd_gan_amd import _lib
L = _lib.lib()
s1, s2 = torch.tensor([0.7], device="cuda"), torch.tensor([1.3], device="cuda")
for (B, IH, N, Cc) in ((64, 8, 512, 512), (64, 4, 512, 512), (32, 8, 512, 512)):
    OH = IH // 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, IH, IH, Cc, generator=g).cuda()
    gy = torch.randn(B, OH, OH, N, generator=g).cuda()
    dw = torch.empty(N, Cc, 4, 4, device="cuda"); db = torch.empty(N, device="cuda")
    geom = K.geom_fwd(B, IH, IH, 4, 2, 1)
    half = (s1, s2, (B // 2) * OH * OH)
    call = lambda: K.wgrad(gy, x, geom, N, Cc, dw, Cc * 16, 16, db=db, half=half)
    call(); torch.cuda.synchronize()
    ref = dw.clone()
    row = f"B={B} {IH}x{IH}->{OH}x{OH} M={B * OH * OH:5d}: plan {timed(call):6.1f} us"
    for cfg in (0, 1, 2, 3, 4, 5, 6, 13):
        for ns in ((-1, 2, 4) if cfg != 13 else (-1, 1, 2, 4, 8)):
            L.mtd_conv_wgrad_override(cfg, ns)
            try:
                t = timed(call)
                torch.cuda.synchronize()
                err = ((dw - ref).abs().max() / ref.abs().max()).item()
                row += f" | cfg{cfg} S={ns}: {t:5.1f}" + ("" if err < 1e-4 else f" (err {err:.1e})")
            except Exception as e:
                row += f" | cfg{cfg} S={ns}: refused"
            finally:
                L.mtd_conv_wgrad_override(-1, -1)
    print(row, flush=True)

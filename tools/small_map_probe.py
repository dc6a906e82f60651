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

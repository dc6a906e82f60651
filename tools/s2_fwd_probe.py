"""Times the 4x4 stride-2 forward convs (arch/Ours/networks.py down{l}) of the step's shapes under every implicit-GEMM tile
configuration and split of K (mtd_conv_igemm_override) -- us per launch, hot inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtd_gan_amd import kernels as K
SHAPES = [(64, 64, 64, 64), (64, 128, 128, 32), (64, 256, 256, 16), (64, 512, 512, 8), (32, 64, 64, 64), (32, 128, 128, 32), (32, 256, 256, 16), (32, 512, 512, 8)]
for (B, Cc, N, H) in SHAPES:
    x = torch.randn(B, H, H, Cc, device="cuda")
    w = torch.randn(N, Cc, 4, 4, device="cuda") * (16 * Cc) ** -0.5
    out = torch.empty(B, H // 2, H // 2, N, device="cuda")
    g = K.geom_fwd(B, H, H, 4, 2, 1)
    bias = torch.randn(N, device="cuda")
    row = f"M{B * H * H // 4:7d} N{N:4d} C{Cc:4d}"
    best = None
    for cfg, sk in [(-1, -1)] + [(c, s) for c in (0, 1, 2, 3, 4) for s in (1, 2, 4)]:
        K.igemm_override(cfg, sk)
        try:
            for _ in range(3):
                K.conv(x, w, g, N, Cc, Cc * 16, 16, out, bias=bias, act=K.ACT_LRELU)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.conv(x, w, g, N, Cc, Cc * 16, 16, out, bias=bias, act=K.ACT_LRELU)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            if cfg < 0:
                row += f"  plan {us:6.1f} |"
            elif best is None or us < best[0]:
                best = (us, cfg, sk)
            if cfg >= 0:
                row += f" {cfg}/{sk}:{us:5.0f}"
        except Exception as e:
            row += f" {cfg}/{sk}: fail"
    K.igemm_override(-1, -1)
    print(row, " best", best, flush=True)

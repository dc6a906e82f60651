"""Times the four-class stride-2 data gradient (conv_multi: arch/Ours/networks.py down{l}, Conv2d(k4, s2, p1) backward w.r.t. its input) of the
step's shapes under every implicit-GEMM tile configuration (mtd_conv_igemm_override) -- us per launch, hot inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtd_gan_amd import kernels as K
SHAPES = [(64, 64, 64, 32), (64, 128, 128, 16), (64, 256, 256, 8), (64, 512, 512, 4), (64, 512, 512, 2), (32, 64, 64, 32), (32, 128, 128, 16), (32, 256, 256, 8), (32, 512, 512, 4)]      # (B, channels in = out of the down conv, ..., output map side of dX / 2)
for (B, Cc, N, H) in SHAPES:
    # conv: x (B, 2H, 2H, Cc) -> y (B, H, H, N); the data gradient: gy (B, H, H, N) -> gx (B, 2H, 2H, Cc)
    gy = torch.randn(B, H, H, N, device="cuda")
    w = torch.randn(N, Cc, 4, 4, device="cuda") * (16 * Cc) ** -0.5
    gx = torch.empty(B, 2 * H, 2 * H, Cc, device="cuda")
    a1, mk = torch.randn_like(gx), torch.randn_like(gx)
    calls = []
    for py in range(2):
        for px in range(2):
            g = K.geom_dgrad_s2(B, 2 * H, 2 * H, py, px)
            calls.append(((gy, w, g, Cc, N, 16, Cc * 16, gx), dict(add1=a1, mask=mk, mask_slope=0.2)))
    row = f"M{B * H * H:7d} x4  N{Cc:4d} C{N:4d}"
    for cfg in (-1, 0, 1, 2, 3, 4, 5):
        K.igemm_override(cfg, -1 if cfg < 0 else 1)
        try:
            for _ in range(3):
                K.conv_multi(calls)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.conv_multi(calls)
            e1.record()
            torch.cuda.synchronize()
            row += f"  cfg{cfg:2d} {1e3 * e0.elapsed_time(e1) / 20:7.1f}"
        except Exception as e:
            row += f"  cfg{cfg:2d}    fail"
    K.igemm_override(-1, -1)
    print(row, flush=True)

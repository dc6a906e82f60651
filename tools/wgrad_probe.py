#!/usr/bin/env python3
"""Lab: weight-gradient kernels alone on the step's 3x3 layer shapes (hot inputs, back-to-back launches incl. their slab reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import _lib, kernels as K

SHAPES = [(32, 64, 64, 64), (32, 32, 128, 128), (32, 16, 256, 256), (32, 8, 512, 512), (32, 8, 1024, 256), (32, 16, 512, 128), (32, 32, 256, 64), (32, 32, 64, 128),
          (64, 16, 256, 256), (64, 64, 64, 64)]
L = _lib.lib()
L.mtd_conv_wgrad_override.argtypes = [_lib.C.c_int, _lib.C.c_int]
for (B, H, C, N) in SHAPES:
    x = torch.randn(B, H, H, C, device="cuda")
    gy = torch.randn(B, H, H, N, device="cuda")
    dw, db = torch.empty(N, C, 3, 3, device="cuda"), torch.empty(N, device="cuda")
    g = K.geom_fwd(B, H, H, 3, 1, 1)
    res = {}
    for name, cfg in (("wino", 16), ("direct", -1 if os.environ.get("MTD_WGRAD_WINO") == "0" else -2)):
        L.mtd_conv_wgrad_override(cfg, -1)
        for _ in range(3):
            K.wgrad(gy, x, g, N, C, dw, C * 9, 9, db=db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.wgrad(gy, x, g, N, C, dw, C * 9, 9, db=db)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    L.mtd_conv_wgrad_override(-1, -1)
    fl = 2.0 * B * H * H * N * C * 9
    print(f"M{B * H * H:7d} N{N:5d} C{C:5d}  winograd {res['wino']:7.1f} us ({fl / res['wino'] / 1e6 * 4 / 9:6.1f} TF executed)   LDS-staged direct {res['direct']:7.1f} us ({fl / res['direct'] / 1e6:6.1f} TF)", flush=True)

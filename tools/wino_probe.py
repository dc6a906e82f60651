#!/usr/bin/env python3
"""Lab: the Winograd conv kernel alone on the step's layer shapes (hot inputs, back-to-back launches, event-timed).
usage: tools/wino_probe.py [B H C N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import kernels as K

SHAPES = [(64, 64, 64, 64), (64, 32, 128, 128), (64, 32, 64, 256), (64, 16, 256, 256), (64, 16, 128, 512), (64, 8, 512, 512), (64, 8, 256, 1024), (64, 32, 256, 64),
          (32, 64, 64, 64), (32, 16, 256, 256)]
if len(sys.argv) > 4:
    v = list(map(int, sys.argv[1:]))
    SHAPES = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]
for (B, H, C, N) in SHAPES:
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(N, C, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, H, H, N, device="cuda")
    g = K.geom_fwd(B, H, H, 3, 1, 1)
    res = {}
    kw = dict(act=K.ACT_LRELU)
    if os.environ.get("EPI", "0") == "1":      # the data gradients' epilogue: two adds, a mask, two scales by batch half
        kw.update(add1=torch.randn_like(out), add2=torch.randn_like(out), mask=torch.randn_like(out), mask_slope=0.2,
                  scale=torch.ones(1, device="cuda"), scale2=torch.ones(1, device="cuda"), scale_split=B * H * H // 2,
                  bias=torch.randn(N, device="cuda"))
    for wino in (True, False):
        K.WINOGRAD = wino
        for _ in range(3):
            K.conv(x, w, g, N, C, C * 9, 9, out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.conv(x, w, g, N, C, C * 9, 9, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[wino] = e0.elapsed_time(e1) / 20 * 1e3
    K.WINOGRAD = True
    fl = 2.0 * B * H * H * N * C * 9
    print(f"M{B * H * H:7d} N{N:5d} C{C:5d}  winograd {res[True]:7.1f} us ({fl / res[True] / 1e6 * 4 / 9:6.1f} TF executed)   direct {res[False]:7.1f} us ({fl / res[False] / 1e6:6.1f} TF)", flush=True)

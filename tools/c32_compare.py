"""Times the two generator-shaped implicit-GEMM kernels (9 = persistent, 10 = halo tile) on random data, B=32, 64x64, 32->32,
with the epilogues the generator uses.  python tools/c32_compare.py"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("mtd-gan_amd.kernels")
_lib = importlib.import_module("mtd-gan_amd._lib")

L = _lib.lib()
L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
B, H = 32, 64
dev = torch.device("cuda")
x = torch.randn(B, H, H, 32, device=dev)
w = torch.randn(32, 32, 3, 3, device=dev) * 0.06
bias = torch.randn(32, device=dev)
res = torch.randn(B, H, H, 32, device=dev)
msk = torch.randn(B, H, H, 32, device=dev)
out = torch.empty(B, H, H, 32, device=dev)
cases = {"bias+lrelu": dict(bias=bias, act=_lib.ACT_LRELU), "bias+add1": dict(bias=bias, add1=res), "plain": {},
         "mask+add1 (dgrad)": dict(mask=msk, mask_slope=0.2, add1=res)}
for name, kw in cases.items():
    row = []
    for cfg in (9, 10):
        L.mtd_conv_igemm_override(cfg, 1)
        g = K.geom_dgrad_s1(B, H, H, 3, 1) if "dgrad" in name else K.geom_fwd(B, H, H, 3, 1, 1)
        wsn, wsc = (9, 32 * 9) if "dgrad" in name else (32 * 9, 9)
        for _ in range(5):
            K.conv(x, w, g, 32, 32, wsn, wsc, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            K.conv(x, w, g, 32, 32, wsn, wsc, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3 / 200)
    print(f"{name:22s} persistent {row[0]:6.1f} us   halo-tile {row[1]:6.1f} us")
L.mtd_conv_igemm_override(-1, -1)

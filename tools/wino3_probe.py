"""Times single Winograd conv launches of the step's shapes on the split-bf16 kernel and on the fp32 MFMA kernel (us per launch,
hot inputs).  `MTD_LAB=1` picks up libmtdgan_hip_lab.so when it exists (tools/wino3_variants.sh builds W3_SKIP variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mtd_gan_amd import _lib, kernels as K
L = _lib.lib()
SHAPES = [(64, 64, 64, 64, 64), (64, 128, 128, 32, 32), (64, 256, 256, 16, 16), (64, 512, 512, 8, 8), (64, 1024, 256, 8, 8), (64, 512, 512, 4, 4), (64, 512, 512, 2, 2)]
pipes = [1] if os.environ.get("W3_ONLY") else [1, 0]
COLD = os.environ.get("W3_COLD") == "1"      # flush the caches between launches (a 1 GB write) and time each launch by itself
DGRAD = os.environ.get("W3_DGRAD") == "1"    # the data gradient's epilogue: two adds and a mask
flush = torch.empty(1 << 28, device="cuda") if COLD else None
print("lab build" if L.mtd_lab_build() else "shipped build", _lib.LIB_PATH)
for (B, Ci, Co, H, W) in SHAPES:
    x = torch.randn(B, H, W, Ci, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (9 * Ci) ** -0.5
    out = torch.empty(B, H, W, Co, device="cuda")
    geom = K.geom_fwd(B, H, W, 3, 1, 1)
    row = f"M{B * H * W:7d} N{Co:5d} C{Ci:5d}"
    kw = dict(act=K.ACT_LRELU)
    if DGRAD:
        kw = dict(add1=torch.randn_like(out), add2=torch.randn_like(out), mask=torch.randn_like(out), mask_slope=0.2)
    for split in pipes:
        L.mtd_set_option(b"wino_split", split)
        K._wino_px_cache.clear(); K._igemm_ws_cache.clear(); K.weights_changed(None)
        for _ in range(3):
            K.conv(x, w, geom, Co, Ci, Ci * 9, 9, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if COLD:
            tot, n = 0.0, 8
            for _ in range(n):
                flush.fill_(1.0)
                e0.record()
                K.conv(x, w, geom, Co, Ci, Ci * 9, 9, out, **kw)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            us = 1e3 * tot / n
        else:
            n = 20
            e0.record()
            for _ in range(n):
                K.conv(x, w, geom, Co, Ci, Ci * 9, 9, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / n
        row += f"   {'bf16x3' if split else 'fp32  '} {us:7.1f} us"
    print(row, flush=True)

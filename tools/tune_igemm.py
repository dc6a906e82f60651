#!/usr/bin/env python3
"""Census + tile/split-K sweep of every distinct implicit-GEMM conv shape of one full training step.

  python tools/tune_igemm.py [--batch 32] [--sweep]

Records the argument structs of one eager step, then times each distinct shape standalone (own buffers,
no concurrency) with the library's automatic plan and -- with --sweep -- under every (tile config, split-K)
override.  Prints one line per shape; used to derive make_plan() in conv_igemm.hip."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K
from mtd_gan_amd.train_step import FullStepWorkload

CFG_BM = [256, 128, 256, 64, 128, 32, 128, 256, 128]
CFG_BN = [32, 32, 64, 64, 128, 128, 32, 32, 128]


def time_call(L, a, iters=20):
    s = K.stream_ptr()
    for _ in range(3):
        _lib.check(L.mtd_conv_igemm(C.byref(a), s), "igemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.mtd_conv_igemm(C.byref(a), s)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    wl = FullStepWorkload(dev, 0, 1, args.batch)
    wl.step_eager()
    torch.cuda.synchronize()
    K.CALL_LOG = []
    wl.step_eager()
    torch.cuda.synchronize()
    log, K.CALL_LOG = K.CALL_LOG, None
    del wl
    L = _lib.lib()
    L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
    shapes = {}
    for kind, raw in log:
        if kind != "igemm":
            continue
        a = _lib.ConvArgs.from_buffer_copy(raw)
        g = a.g
        key = (bytes(g), a.N, a.C, a.in_ld, a.out_ld, a.w_sn, a.w_st, bool(a.add1), bool(a.add2), bool(a.mask), bool(a.bias), bool(a.scale), a.act)
        shapes.setdefault(key, [0, raw])[0] += 1
    print(f"{len(shapes)} distinct igemm shapes, {sum(v[0] for v in shapes.values())} launches per step")
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    rows = []
    for key, (count, raw) in shapes.items():
        a = _lib.ConvArgs.from_buffer_copy(raw)
        g = a.g
        T = g.TH * g.TW
        M = g.B * g.OH * g.OW
        xin = torch.randn(g.B * g.IH * g.IW * a.in_ld, device=dev)
        wsz = (a.N - 1) * a.w_sn + a.C + 16 * a.w_st + 64
        w = torch.randn(wsz, device=dev) * 0.05
        npix_out = g.B * g.OHF * g.OWF
        out = torch.empty(npix_out * a.out_ld, device=dev)
        a.inp, a.w, a.out = xin.data_ptr(), w.data_ptr(), out.data_ptr()
        keep = []
        for f, ldf in (("add1", "add1_ld"), ("add2", "add2_ld"), ("mask", "mask_ld")):
            if getattr(a, f):
                t = torch.randn(npix_out * getattr(a, ldf), device=dev)
                keep.append(t)
                setattr(a, f, t.data_ptr())
        for f in ("bias", "scale"):
            if getattr(a, f):
                t = torch.ones(a.N, device=dev)
                keep.append(t)
                setattr(a, f, t.data_ptr())
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        flops = 2.0 * M * a.N * a.C * T
        L.mtd_conv_igemm_override(-1, -1)
        t_auto = time_call(L, a)
        best = (t_auto, "auto")
        table = []
        if args.sweep:
            chunks = a.C // 32
            for cfg in range(9):
                if a.N % CFG_BN[cfg] or (cfg in (6, 7) and T > 9):
                    continue
                blocks = ((M + CFG_BM[cfg] - 1) // CFG_BM[cfg]) * (a.N // CFG_BN[cfg])
                if blocks > 4096 and CFG_BM[cfg] < 128:
                    continue
                for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                    if sk > chunks or (sk > 1 and blocks * sk > 2048):
                        continue
                    if sk * M * a.N * 4 > ws.numel():
                        continue
                    L.mtd_conv_igemm_override(cfg, sk)
                    t = time_call(L, a, iters=10)
                    table.append((t, cfg, sk))
                    if t < best[0]:
                        best = (t, f"cfg{cfg}/s{sk}")
            L.mtd_conv_igemm_override(-1, -1)
        rows.append((count * t_auto, count, M, a.N, a.C, T, g.in_sy, t_auto, flops / t_auto / 1e6, best, sorted(table)[:4]))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    tot_best = sum(r[1] * r[9][0] for r in rows)
    print(f"sum of standalone igemm time per step: auto {tot / 1e3:.2f} ms, best {tot_best / 1e3:.2f} ms")
    for r in rows:
        print(f"x{r[1]:3d} M={r[2]:6d} N={r[3]:4d} C={r[4]:4d} T={r[5]:2d} s{r[6]}  auto {r[7]:7.1f} us {r[8]:6.1f} TF  tot {r[0] / 1e3:6.2f} ms"
              f"  best {r[9][1]} {r[9][0]:7.1f} us  top: {[(round(t, 1), c, s) for t, c, s in r[10]]}")


if __name__ == "__main__":
    main()

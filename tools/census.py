#!/usr/bin/env python3
"""Standalone timing of every distinct weight-gradient and vector-ALU conv call of one training step.

  python tools/census.py [--batch 32]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K
from mtd_gan_amd.train_step import FullStepWorkload


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--sweep-wgrad", action="store_true", help="sweep (tile/tap-group config, pixel splits) per MFMA wgrad shape")
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
    ws = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
    s = K.stream_ptr()
    wg, dr = {}, {}
    for kind, raw in log:
        if kind == "wgrad":
            a = _lib.WgradArgs.from_buffer_copy(raw)
            key = (bytes(a.g), a.N, a.C, a.p_ld, a.q_ld, a.w_sn, a.w_sc, bool(a.db), a.accumulate)
            wg.setdefault(key, [0, raw])[0] += 1
        elif kind == "direct":
            a = _lib.ConvArgs.from_buffer_copy(raw)
            key = (bytes(a.g), a.N, a.C, a.in_ld, a.out_ld, a.w_sn, a.w_sc, a.w_st, bool(a.add1), bool(a.mask), bool(a.bias), a.act)
            dr.setdefault(key, [0, raw])[0] += 1
    rows = []
    tbest = [0.0]
    for key, (count, raw) in wg.items():
        a = _lib.WgradArgs.from_buffer_copy(raw)
        g = a.g
        T = g.TH * g.TW
        M = g.B * g.OH * g.OW
        pt = torch.randn(M * a.p_ld, device=dev)
        qt = torch.randn(g.B * g.IH * g.IW * a.q_ld, device=dev)
        dw = torch.zeros(max(a.N * a.w_sn, a.C * a.w_sc) + g.KW * g.KW * 4 + 64, device=dev)
        db = torch.zeros(a.N, device=dev)
        a.p, a.q, a.dw = pt.data_ptr(), qt.data_ptr(), dw.data_ptr()
        if a.db:
            a.db = db.data_ptr()
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        t = timed(lambda: _lib.check(L.mtd_conv_wgrad(C.byref(a), s), "wgrad"))
        fl = 2.0 * M * a.N * a.C * T
        extra = ""
        if args.sweep_wgrad and a.N % 32 == 0 and a.C % 32 == 0:
            L.mtd_conv_wgrad_override.argtypes = [C.c_int, C.c_int]
            WN = [1, 1, 2, 1, 1, 1, 2]
            TG = [9, 4, 1, 8, 3, 1, 3]
            res = []
            for cfg in range(7):
                if a.N % (32 * WN[cfg]) or a.C % (32 * WN[cfg]):
                    continue
                if T > 1 and TG[cfg] == 1 and WN[cfg] == 1 and a.N * a.C >= 128 * 128:
                    continue
                ntg = (T + TG[cfg] - 1) // TG[cfg]
                tiles = (a.N // (32 * WN[cfg])) * (a.C // (32 * WN[cfg])) * ntg
                for target in (128, 256, 512, 1024, 2048):
                    ns = max(1, (target + tiles - 1) // tiles)
                    if ns > (M + 127) // 128:
                        ns = (M + 127) // 128
                    L.mtd_conv_wgrad_override(cfg, ns)
                    need = L.mtd_conv_wgrad_ws_bytes(C.byref(a))
                    if need == 0 or need > ws.numel():
                        continue
                    tt = timed(lambda: L.mtd_conv_wgrad(C.byref(a), s), iters=5)
                    res.append((round(tt, 1), cfg, ns))
            L.mtd_conv_wgrad_override(-1, -1)
            res = sorted(set(res))
            if res:
                extra = f"  best {res[0]}  next {res[1:4]}"
                tbest[0] += count * min(res[0][0], t)
        elif args.sweep_wgrad:
            tbest[0] += count * t
        rows.append((count * t, f"wgrad  x{count:3d} M={M:6d} N={a.N:4d} C={a.C:4d} T={T:2d} s{g.in_sy} d{g.tap_dy}  {t:7.1f} us {fl / t / 1e6:6.1f} TF  tot {count * t / 1e3:6.2f} ms" + extra))
    for key, (count, raw) in dr.items():
        a = _lib.ConvArgs.from_buffer_copy(raw)
        g = a.g
        T = g.TH * g.TW
        M = g.B * g.OH * g.OW
        xin = torch.randn(g.B * g.IH * g.IW * a.in_ld, device=dev)
        w = torch.randn((a.N - 1) * a.w_sn + (a.C - 1) * a.w_sc + 16 * max(a.w_st, 1) + 64, device=dev)
        npix = g.B * g.OHF * g.OWF
        out = torch.empty(npix * a.out_ld, device=dev)
        a.inp, a.w, a.out = xin.data_ptr(), w.data_ptr(), out.data_ptr()
        keep = []
        for f, ldf in (("add1", "add1_ld"), ("add2", "add2_ld"), ("mask", "mask_ld")):
            if getattr(a, f):
                tt = torch.randn(npix * getattr(a, ldf), device=dev)
                keep.append(tt)
                setattr(a, f, tt.data_ptr())
        for f in ("bias", "scale"):
            if getattr(a, f):
                tt = torch.ones(a.N, device=dev)
                keep.append(tt)
                setattr(a, f, tt.data_ptr())
        t = timed(lambda: _lib.check(L.mtd_conv_direct(C.byref(a), s), "direct"))
        byts = 4.0 * (g.B * g.IH * g.IW * a.C + M * a.N)
        rows.append((count * t, f"direct x{count:3d} M={M:6d} N={a.N:4d} C={a.C:4d} T={T:2d} s{g.in_sy} in_ld={a.in_ld} out_ld={a.out_ld}  {t:7.1f} us {byts / t / 1e3:7.1f} GB/s  tot {count * t / 1e3:6.2f} ms"))
    rows.sort(key=lambda r: -r[0])
    print(f"wgrad total {sum(r[0] for r in rows if r[1].startswith('wgrad')) / 1e3:.2f} ms/step, direct total {sum(r[0] for r in rows if r[1].startswith('direct')) / 1e3:.2f} ms/step")
    if args.sweep_wgrad:
        print(f"wgrad with the best swept plan per shape: {tbest[0] / 1e3:.2f} ms/step")
    for r in rows:
        print(r[1])


if __name__ == "__main__":
    main()

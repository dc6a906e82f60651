"""Times the fused data + weight gradient launch of a generator layer (csrc/conv_c32_bwd.hip) against the two launches it
replaces, B = 32.  MTD_C32F_ROLES=1 / 2: only one role of the fused kernel computes (the other only joins the barriers).
python tools/c32f_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: F401,E402
from mtd_gan_amd import kernels as K  # noqa: E402

B, H = 32, 64
dev = torch.device("cuda")
r = lambda *s: torch.randn(*s, device=dev)
p_, q_, add1, mask = r(B, H, H, 32), r(B, H, H, 32), r(B, H, H, 32), r(B, H, H, 32)
w = r(32, 32, 3, 3) * 0.1
out, out2 = torch.empty(B, H, H, 32, device=dev), torch.empty(B, H, H, 32, device=dev)
dw, db = torch.zeros(32, 32, 3, 3, device=dev), torch.zeros(32, device=dev)
gf, gt = K.geom_fwd(B, H, H, 3, 1, 1), K.geom_dgrad_s1(B, H, H, 3, 1)
dg = ((p_, w, gt, 32, 32, 9, 32 * 9, out), dict(add1=add1, mask=mask, mask_slope=0.0, out2=out2))
wg = ((p_, q_, gf, 32, 32, dw, 32 * 9, 9), dict(db=db))


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def two():
    d = K.DeferredWgrads()
    K.wgrad(*wg[0], db=db, defer=d)
    K.conv(*dg[0], **dg[1])


def conv_only():
    K.conv(*dg[0], **dg[1])


def wgrad_only():
    d = K.DeferredWgrads()
    K.wgrad(*wg[0], db=db, defer=d)


def fused():
    d = K.DeferredWgrads()
    assert K.conv_wgrad_fused(dg, wg, d)


print(f"roles={os.environ.get('MTD_C32F_ROLES', '3')}  dgrad {timeit(conv_only):6.1f} us  wgrad {timeit(wgrad_only):6.1f} us  "
      f"two launches {timeit(two):6.1f} us  fused {timeit(fused):6.1f} us", flush=True)

if os.environ.get("MTD_C32F_STAMPS"):
    import ctypes as C
    from mtd_gan_amd import _lib
    fused()
    buf = (C.c_ulonglong * 256)()
    L = _lib.lib()
    L.mtd_conv_c32_bwd_stamps.argtypes = [C.c_void_p]
    assert L.mtd_conv_c32_bwd_stamps(C.cast(buf, C.c_void_p)) == 0
    names = {0: "d:start", 1: "d:after B0", 2: "d:blk0", 3: "d:blk1", 4: "d:blk2", 5: "d:blk3", 6: "d:exit", 8: "w:start", 9: "w:after B0", 10: "w:q0", 11: "w:q1",
             12: "w:q2", 13: "w:q3", 14: "w:exit"}
    for wg in (0, 5, 15):
        t0 = buf[wg * 16 + 0]
        print(f"workgroup {wg}: " + "  ".join(f"{names[i]} {(buf[wg * 16 + i] - t0) / 100.0:.1f}" for i in sorted(names) if buf[wg * 16 + i]), "(us at 100 MHz)")

#!/usr/bin/env python3
"""Where is the GPU idle time around descriptor-table uploads?  Events before / after each mtd_upload in a real step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K
from mtd_gan_amd.train_step import FullStepWorkload

dev = torch.device("cuda", 0)
wl = FullStepWorkload(dev, 0, 1, 32)
for _ in range(3):
    wl.step_eager()
torch.cuda.synchronize()
L = _lib.lib()
orig = L.mtd_upload
probes = []


def wrapped(src, dst, nbytes, stream):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    t0 = time.perf_counter_ns()
    rc = orig(src, dst, nbytes, stream)
    e1.record()
    # a second, device-to-device copy of the same size right behind it
    orig(dst, scratch.data_ptr(), nbytes, stream)
    e2.record()
    probes.append((t0, nbytes, e0, e1, e2))
    return rc


scratch = torch.empty(1 << 20, dtype=torch.uint8, device=dev)


class Shim:
    def __getattr__(self, name):
        if name == "mtd_upload":
            return wrapped
        return getattr(L, name)


_lib._lib = Shim()
a_ev = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a_ev.record()
torch.cuda.synchronize()
ta = time.perf_counter_ns()
for _ in range(3):
    wl.step_eager()
torch.cuda.synchronize()
for t0, nbytes, e0, e1, e2 in probes:
    g0 = a_ev.elapsed_time(e0)
    print(f"host {(t0 - ta) / 1e6:8.2f} ms  gpu reaches upload at {g0:8.2f} ms  bytes {nbytes:6d}  pinned upload {e0.elapsed_time(e1) * 1e3:8.1f} us   d2d copy {e1.elapsed_time(e2) * 1e3:8.1f} us")

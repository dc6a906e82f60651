#!/usr/bin/env python3
"""Does a kernel that reads freshly written pinned host memory stall a deeply queued stream?  (diagnostic)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K

dev = torch.device("cuda", 0)
L = _lib.lib()
a = torch.randn(1 << 24, device=dev)          # 64 MB: mul takes ~40 us
b = torch.empty_like(a)
host = torch.zeros(16 << 20, dtype=torch.uint8).pin_memory()
dst = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
pageable = torch.zeros(4096, dtype=torch.uint8)


def run(mode, n=3000, every=50):
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    e_start = torch.cuda.Event(enable_timing=True)
    e_start.record()
    for i in range(n):
        torch.mul(a, 1.0001, out=b)
        if i % every == every - 1:
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            o = ((i // every) * 4096) % (16 << 20)
            if mode == "pinned_fresh":
                host[o:o + 4096].copy_(pageable)
                L.mtd_upload(host.data_ptr() + o, dst.data_ptr() + o, 4096, K.stream_ptr())
            elif mode == "pinned_static":
                L.mtd_upload(host.data_ptr(), dst.data_ptr(), 4096, K.stream_ptr())
            elif mode == "d2d":
                L.mtd_upload(dst.data_ptr(), dst.data_ptr() + 8192, 4096, K.stream_ptr())
            elif mode == "memcpy":
                host[o:o + 4096].copy_(pageable)
                dst[o:o + 4096].copy_(host[o:o + 4096], non_blocking=True)
            eb.record()
            evs.append((ea, eb))
    th = time.perf_counter() - t0
    e_end = torch.cuda.Event(enable_timing=True)
    e_end.record()
    e_end.synchronize()
    d = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    print(f"{mode:14s} total gpu {e_start.elapsed_time(e_end):8.2f} ms  host enqueue {th * 1e3:8.2f} ms   marker->marker us: median {d[len(d) // 2]:7.1f}  max {d[-1]:8.1f}  mean {sum(d) / len(d):7.1f}")


for mode in ("none", "d2d", "pinned_static", "pinned_fresh", "memcpy", "none"):
    run(mode)

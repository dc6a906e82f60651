#!/usr/bin/env python3
"""cProfile of the host side of the generator-only workload (fwd + bwd): where the enqueue time goes, and the pure
enqueue time of a step (no synchronisation inside the loop)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import bench_workloads as BW

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = BW.make("generator", dev, 0, 1, 32)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    wl.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, with drain {1e3 * (t2 - t0) / 20:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    wl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

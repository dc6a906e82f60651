#!/usr/bin/env python3
"""Lab: per-step wall time of 240 eager steps (synchronised every step) with the allocator's and the garbage collector's counters,
to find what an occasional +60 ms step is."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtd_gan_amd  # noqa
from mtd_gan_amd.train_step import FullStepWorkload
wl = FullStepWorkload(torch.device("cuda", 0), 0, 1, 32)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
def snap():
    st = torch.cuda.memory_stats()
    return (st["num_device_alloc"], st["num_device_free"], st["num_alloc_retries"], st["reserved_bytes.all.current"] >> 20,
            tuple(s["collections"] for s in gc.get_stats()))
prev = snap()
for i in range(240):
    t0 = time.perf_counter()
    wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cur = snap()
    if (t2 - t0) > 0.036 or cur[:3] != prev[:3]:
        print(f"step {i}: host {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms, alloc/free/retries/reservedMB/gc {prev} -> {cur}", flush=True)
    prev = cur
print("done")

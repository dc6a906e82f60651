"""How long does the host need to ENQUEUE one step vs how long the GPU needs to run it?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mtd_gan_amd import bench_workloads as BW
for name in sys.argv[1:] or ["generator", "full_step"]:
    wl = BW.make(name, torch.device("cuda"), 0, 1, 32)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
    print(f"{name}: host enqueue {min(enq):.2f} ms (median {sorted(enq)[2]:.2f}), enqueue+drain {min(tot):.2f} ms", flush=True)

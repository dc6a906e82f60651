"""Soak run of the eager training step: memory must stay flat, losses finite, the descriptor arena must survive its
recycling.  python tools/soak.py [--steps 400]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import kernels as K
from mtd_gan_amd.train_step import FullStepWorkload

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--fresh-inputs", action="store_true", help="new input tensors every step (new descriptor tables: exercises the arena recycling)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = FullStepWorkload(dev, 0, 1, 32)
t0 = time.perf_counter()
keep = []
for i in range(a.steps):
    if a.fresh_inputs:
        keep.append((wl.x, wl.y))                 # hold a few old batches so that the allocator hands out new addresses
        keep = keep[-7:]
        wl.x, wl.y = wl.x.clone(), wl.y.clone()
    wl.step()
    if i % 50 == 0 or i == a.steps - 1:
        torch.cuda.synchronize()
        fin = all(torch.isfinite(p).all().item() for p in wl.model.parameters())
        print(f"step {i:4d}  allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  "
              f"arena ofs {K.arena(dev).ofs}  tables {len(K._desc_cache)}  finite {fin}  {time.perf_counter() - t0:6.1f} s", flush=True)

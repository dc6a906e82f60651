#!/usr/bin/env python3
"""cProfile of the host side of the eager full training step (where does the enqueue time go)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd.train_step import FullStepWorkload

dev = torch.device("cuda", 0)
wl = FullStepWorkload(dev, 0, 1, 32)
for _ in range(3):
    wl.step_eager()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    wl.step_eager()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(45)

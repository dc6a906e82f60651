#!/usr/bin/env python3
"""Where does the host spend its enqueue time?  cProfile of five eager training steps (the GPU runs behind: the profile is the
Python + ctypes + HIP-runtime launch path, not kernel time)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mtd_gan_amd  # noqa: F401
from mtd_gan_amd.train_step import FullStepWorkload

wl = FullStepWorkload(torch.device("cuda", 0), 0, 1, 32)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    wl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))

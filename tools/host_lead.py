#!/usr/bin/env python3
"""How far does the host run ahead of the GPU at the phase boundaries of the eager training step?
Records (host time, HIP event) pairs at the boundaries of 6 consecutive steps after one anchor synchronisation;
lead = GPU time - host time at the same marker (positive: the GPU is the bottleneck there)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd.train_step import FullStepWorkload

dev = torch.device("cuda", 0)
wl = FullStepWorkload(dev, 0, 1, 32)
m, oG, oD, wm = wl.model, wl.oG, wl.oD, wl.wm
D, G = m.Discriminator, m.Generator
marks = []


def mark(label):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((label, time.perf_counter_ns(), ev))


def step():
    mark("step start")
    oD.zero_grad(); D.zero_grad()
    d_losses, d_details = m.d_loss(wl.x, wl.y)
    mark("d_loss fwd done")
    wm.backward(losses=d_losses, shared_parameters=list(D.shared_parameters()),
                task_specific_parameters=list(D.task_specific_parameters()), last_shared_parameters=list(D.last_shared_parameters()))
    mark("D backward+pcgrad done")
    oD.step()
    mark("D adamw done")
    oG.zero_grad(); G.zero_grad()
    g_loss, g_details = m.g_loss(wl.x, wl.y)
    mark("g_loss fwd done")
    g_loss.backward()
    mark("G backward done")
    oG.step()
    mark("G adamw done")


for _ in range(3):
    step()
torch.cuda.synchronize()
marks.clear()
a_ev = torch.cuda.Event(enable_timing=True)
a_ev.record()
torch.cuda.synchronize()
t0 = time.perf_counter_ns()
for _ in range(6):
    step()
torch.cuda.synchronize()
prev_h = prev_g = 0.0
for label, th, ev in marks:
    h = (th - t0) / 1e6
    g = a_ev.elapsed_time(ev)
    print(f"{label:26s} host {h:8.2f} ms (+{h - prev_h:6.2f})   gpu {g:8.2f} ms (+{g - prev_g:6.2f})   lead {g - h:7.2f} ms")
    prev_h, prev_g = h, g
starts = [(th - t0) / 1e6 for label, th, ev in marks if label == "step start"]
ends = [(th - t0) / 1e6 for label, th, ev in marks if label == "G adamw done"]
gstart = [a_ev.elapsed_time(ev) for label, th, ev in marks if label == "step start"]
gend = [a_ev.elapsed_time(ev) for label, th, ev in marks if label == "G adamw done"]
host_ms = [e - s for s, e in zip(starts, ends)]
print(f"host enqueue per step: {sum(host_ms) / len(host_ms):.2f} ms (min {min(host_ms):.2f}, max {max(host_ms):.2f}); "
      f"GPU per step: {(gend[-1] - gstart[0]) / len(gstart):.2f} ms; host / GPU = {sum(host_ms) / len(host_ms) / ((gend[-1] - gstart[0]) / len(gstart)):.2f} "
      f"(pid {os.getpid()}, {os.cpu_count()} CPUs visible)")

#!/usr/bin/env python3
"""How much host time does one training iteration cost, and how far does the host run ahead of the GPU?

  python tools/host_lead.py [--mode list|eager] [--steps N] [--procs P]

eager: (host time, HIP event) pairs at the phase boundaries of consecutive eager steps after one anchor synchronisation;
       lead = GPU time - host time at the same marker (positive: the GPU is the bottleneck there).
list : the recorded launch list (train_step.RecordedTrainStep, the default of engine.train_MTD_GAN_Ours): host time inside
       RecordedTrainStep.step() per iteration (everything the host does for an iteration: input copies, the pinned-slot
       refresh, the replay loop) against the GPU time per iteration.
--procs P: P concurrent copies of this measurement on this box (the GPU is shared, so the GPU time per iteration grows P-fold;
       the HOST time per iteration is what is being checked: do P ranks' Python loops on one host get in each other's way?)."""
import argparse
import os
import subprocess
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="list")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--procs", type=int, default=1)
args = ap.parse_args()

if args.procs > 1:
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", args.mode, "--steps", str(args.steps)]
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(args.procs)]
    outs = [p.communicate()[0] for p in ps]
    for i, o in enumerate(outs):
        for ln in [l for l in o.splitlines() if l.startswith(("list replay", "  replay into", "host enqueue"))] or ["<no output>"]:
            print(f"[proc {i}] " + ln)
    sys.exit(max(p.returncode for p in ps))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import train_step as TS
from mtd_gan_amd.train_step import FullStepWorkload

dev = torch.device("cuda", 0)
TS.LIST_MODE = args.mode == "list"
wl = FullStepWorkload(dev, 0, 1, 32)

if args.mode == "list":
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    st = wl.model._mtd_recorded
    assert isinstance(st, TS.RecordedTrainStep), st
    host, cpu = [], []
    real = st.step

    def timed_step(x=None, y=None):
        t0, c0 = time.perf_counter_ns(), time.thread_time_ns()
        out = real(x, y)
        host.append((time.perf_counter_ns() - t0) / 1e6)
        cpu.append((time.thread_time_ns() - c0) / 1e6)          # CPU time of this thread: without the waits on a full hardware queue
        return out
    st.step = timed_step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / args.steps
    host.sort()
    n_launch = sum(1 for f, _a in st.list.ops if getattr(f, "__name__", "").startswith("mtd_"))
    print(f"list replay: host {sum(host) / len(host):.2f} ms wall inside step() per iteration (median {host[len(host) // 2]:.2f}, max {host[-1]:.2f}; "
          f"includes waiting on the full hardware queue), {sum(cpu) / len(cpu):.2f} ms CPU time of the thread; "
          f"iteration {wall:.2f} ms wall; host / iteration = {sum(host) / len(host) / wall:.2f}; {len(st.list.ops)} recorded operations, "
          f"{n_launch} library launches (pid {os.getpid()}, {os.cpu_count()} CPUs visible)")
    # the same with an EMPTY queue in front of every replay (synchronise first): what the host spends on an iteration when no
    # back-pressure from a full hardware queue is in it
    st.step = real
    cold = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter_ns()
        st.step()
        cold.append((time.perf_counter_ns() - t0) / 1e6)
    torch.cuda.synchronize()
    print(f"  replay into an empty queue: {min(cold):.2f} - {max(cold):.2f} ms host")
    # where the replay loop spends it, by entry point (one replay, timed per operation: adds ~0.1 us per operation)
    by = {}
    torch.cuda.synchronize()
    for f, a in st.list.ops:
        t0 = time.perf_counter_ns()
        f(*a)
        dt = time.perf_counter_ns() - t0
        nme = getattr(f, "__name__", None) or type(f).__name__
        e = by.setdefault(nme, [0, 0])
        e[0] += dt
        e[1] += 1
    torch.cuda.synchronize()
    for nme, (ns, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {nme:34s} {n:5d} calls  {ns / 1e6:7.3f} ms  {ns / 1e3 / n:7.2f} us each")
    sys.exit(0)

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

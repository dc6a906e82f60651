#!/usr/bin/env python3
"""Per-shape table of the full step's MFMA launches (library launch profiler, one stream): kernel, M, N, C, taps,
launches per step, us per launch, TFLOP/s, ms per step.  Says which layers the next kernel revision should look at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401
from mtd_gan_amd import bench_workloads as BW, kernels as K

dev = torch.device("cuda", 0)
wl = BW.make(os.environ.get("WL", "full_step"), dev, 0, 1, int(os.environ.get("BATCH", "32")))
for _ in range(3):
    wl.step()
K.set_concurrency(False)
step = getattr(wl, "step_eager", wl.step)
step()
steps = 5
cap = 8192 * steps
K.prof_enable(cap)
for _ in range(steps):
    step()
torch.cuda.synchronize()
recs = K.prof_collect(cap)
K.prof_enable(0)
K.set_concurrency(True)
by = {}
for r in recs:
    key = (r["kernel"], r["M"], r["N"], r["C"], r["taps"], r["splitk"])
    d = by.setdefault(key, [0.0, 0.0, 0])
    d[0] += r["ms"]; d[1] += r["flops"]; d[2] += 1
rows = sorted(by.items(), key=lambda kv: -kv[1][0])
print(f"{'kernel':34s} {'M':>8s} {'N':>4s} {'C':>4s} {'T':>3s} {'S':>3s} {'n/step':>6s} {'us':>8s} {'TF/s':>7s} {'ms/step':>8s}")
tot = 0.0
for (k, M, N, Cc, T, S), (ms, fl, n) in rows[:int(os.environ.get("TOP", "45"))]:
    print(f"{k:34s} {M:8d} {N:4d} {Cc:4d} {T:3d} {S:3d} {n / steps:6.1f} {1e3 * ms / n:8.1f} {fl / ms / 1e9:7.1f} {ms / steps:8.3f}")
print("sum ms/step (event-pair overhead ~5 us per launch included):", round(sum(v[0] for v in by.values()) / steps, 2))

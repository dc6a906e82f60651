"""Wall time of the phases of one training iteration (engine.train_iteration), each closed by a device synchronisation, so
the phases add up to slightly more than the pipelined step.  python tools/phase_times.py [--batch 32]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TS = importlib.import_module("mtd-gan_amd.train_step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    wl = TS.FullStepWorkload(dev, 0, 1, a.batch)
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    model, x, y, oG, oD, wm = wl.model, wl.x, wl.y, wl.oG, wl.oD, wl.wm
    D, G = model.Discriminator, model.Generator
    acc = {}

    def phase(name, fn):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r

    n = 5
    for _ in range(n):
        oD.zero_grad(); D.zero_grad()
        dl, dd = phase("D: G forward + 4 SN + D forward x2 + losses", lambda: model.d_loss(x, y))
        phase("D: 3-task backward + PCGrad", lambda: wm.backward(losses=dl, shared_parameters=list(D.shared_parameters()),
                                                               task_specific_parameters=list(D.task_specific_parameters()),
                                                               last_shared_parameters=list(D.last_shared_parameters())))
        phase("D: AdamW", lambda: oD.step())
        oG.zero_grad(); G.zero_grad()
        gl, gd = phase("G: D forward on fake + losses", lambda: model.g_loss(x, y))
        phase("G: backward through D and G", lambda: gl.backward())
        phase("G: AdamW", lambda: oG.step())
    tot = 0.0
    for k, v in acc.items():
        print(f"{k:50s} {v / n:7.2f} ms")
        tot += v / n
    print(f"{'sum of synchronised phases':50s} {tot:7.2f} ms")


main()

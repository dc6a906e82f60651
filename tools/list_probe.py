#!/usr/bin/env python3
"""Launch-list replay of the generator workload: host time per replay, step time with and without side streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__  # noqa: F401  (registers the package alias)
from mtd_gan_amd import bench_workloads as BW, kernels as K

dev = torch.device("cuda", 0)
for conc in (True, "wgrad-only", False):
    K.set_concurrency(bool(conc))
    if conc == "wgrad-only":                      # weight gradients beside the data-gradient chain, spectral branch inline
        K.side_stream(dev, 1).enabled = False
    os.environ["MTD_GRAPH"] = "list"
    wl = BW.GeneratorWorkload(dev, 0, 1, 32)
    assert wl.launch_list is not None, wl.graph_error
    ll = wl.launch_list
    nk = sum(1 for f, a in ll.ops if hasattr(f, "argtypes"))
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(20):
        h0 = time.perf_counter()
        wl.step()
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    # host alone: replay after a sync, time until the call returns
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    wl.step()
    h1 = time.perf_counter() - h0
    torch.cuda.synchronize()
    print(f"side_streams={conc} ops={len(ll.ops)} launches={nk} step={dt*1e3:.3f} ms host_enqueue(first, idle queue)={h1*1e3:.3f} ms "
          f"host_enqueue(steady, median)={sorted(host)[10]*1e3:.3f} ms", flush=True)
    del wl
K.set_concurrency(True)

#!/usr/bin/env python3
"""Column transforms + channel mix of the Res-FFT-Conv block at the training size (32 patches): us per launch of
mtd_spec_mix_fwd4 / mtd_spec_mix_bwd4 (csrc/resfft4.hip) and of mtd_rfft_rows, each over a ring of operand sets (the previous
launch's operands are not the next one's), and the byte floors they are held against.  Lab switches (MTD_LAB=1 and a lab build)
are read by the library once per process: one process per configuration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mtd_gan_amd  # noqa: F401
from mtd_gan_amd import _lib, kernels as K


def timed(fns, reps=8, iters=5):
    """us per launch inside a replayed hipGraph of reps passes over the ring (the product's own launch mode; no host time)."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            for f in fns:
                f()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (iters * reps * len(fns))


def main():
    B = int(os.environ.get("PROBE_B", "32"))
    dev = torch.device("cuda", 0)
    ring = 6
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.randn(B, 64, 64, 32, device=dev, generator=g) for _ in range(ring)]
    w2 = (torch.randn(64, 64, device=dev, generator=g) * 0.1)
    w2t = w2.t().contiguous()
    b2 = torch.randn(64, device=dev, generator=g) * 0.1
    Rs = [K.rfft_rows(x, False) for x in xs]
    outs = [K.spec_mix_fwd(R, w2t, b2, True) for R in Rs]
    dw2, db2 = torch.zeros(64, 64, device=dev), torch.zeros(64, device=dev)
    t_rows = timed([lambda x=x: K.rfft_rows(x, False) for x in xs])
    t_fwd = timed([lambda R=R: K.spec_mix_fwd(R, w2t, b2, True) for R in Rs])
    t_bwd = timed([lambda R=R, o=o: K.spec_mix_bwd(R, w2, o[1], o[2], dw2, db2) for R, o in zip(Rs, outs)])
    mb = B * 33 * 64 * 64 * 4 / 1e6
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MTD_SPECMIX"))
    print(f"B={B} [{tag}] rfft_rows {t_rows:6.2f} us   spec_mix_fwd {t_fwd:6.2f} us ({3 * mb / t_fwd / 1e3:.2f} TB/s)   "
          f"spec_mix_bwd {t_bwd:6.2f} us ({(3 * mb + mb + 544 * 4224 * 4 / 1e6) / t_bwd / 1e3:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()

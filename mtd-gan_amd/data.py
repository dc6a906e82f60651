"""Synthetic LDCT-shaped patches for benchmarks / smoke runs (no dataset ships with this repo).

Mimics the value distribution produced by the reference's loader (create_datasets/Mayo.py:119-136):
a smooth HU field windowed to [-160, 240] HU -> [0, 1]; the low-dose input adds 40 HU of noise before the
window.  Recipe fixed by SURVEY.md 8d (seed -> identical tensors on every rank/machine)."""
import torch
import torch.nn.functional as F


def synthetic_ldct(batch, seed=1234, size=64):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(batch, 1, size, size, generator=g)
    box = torch.full((1, 1, 5, 5), 1.0 / 25.0)
    for _ in range(3):
        z = F.conv2d(F.pad(z, (2, 2, 2, 2), mode="replicate"), box)
    hu = 150.0 * z / z.std() + 40.0
    noise = torch.randn(batch, 1, size, size, generator=g)
    y = ((hu + 160.0) / 400.0).clamp(0, 1)
    x = ((hu + 40.0 * noise + 160.0) / 400.0).clamp(0, 1)
    return x, y

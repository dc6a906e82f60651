#!/usr/bin/env python3
"""Numerical probe (CPU, fp32 emulation): Winograd F(2x4, 3x3) -- F(2,3) down the rows, F(4,3) along them: 24 multiplications
per 2 x 4 output tile = 3 per output pixel, against 4 for F(2x2, 3x3) and 2.25 for F(4x4, 3x3) -- on discriminator-shaped
layers.  Error of each form in fp32 against the float64 direct convolution, relative to the output's max-abs (the measure of
the GPU parity tests, bound 1e-3), forward and data gradient alike (a data gradient is a convolution with the flipped filter)."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
f64 = torch.float64
# F(2,3)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=f64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=f64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=f64)
# F(4,3), interpolation points 0, +-1, +-2, inf (Lavin & Gray)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=f64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=f64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=f64)


def wino(x, w, BTy, Gy, ATy, BTx, Gx, ATx, dt):
    """x (B,C,H,W), w (N,C,3,3), pad 1, in dtype dt with fp32 accumulation emulated by dt matmuls."""
    my, mx = ATy.shape[0], ATx.shape[0]
    ty, tx = BTy.shape[0], BTx.shape[0]
    B, C, H, W = x.shape
    N = w.shape[0]
    xp = F.pad(x.to(dt), (1, 1 + (-W) % mx, 1, 1 + (-H) % my))
    tiles = xp.unfold(2, ty, my).unfold(3, tx, mx)                       # B,C,nty,ntx,ty,tx
    U = torch.einsum("ai,bcyxij,dj->bcyxad", BTy.to(dt), tiles, BTx.to(dt))
    Wt = torch.einsum("ai,ncij,dj->ncad", Gy.to(dt), w.to(dt), Gx.to(dt))
    Mm = torch.einsum("bcyxad,ncad->bnyxad", U, Wt)
    Y = torch.einsum("pa,bnyxad,qd->bnyxpq", ATy.to(dt), Mm, ATx.to(dt))
    nty, ntx = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, N, nty * my, ntx * mx)[:, :, :H, :W]


def err(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


print(f"{'layer':26s} {'direct fp32':>12s} {'F(2x2) fp32':>12s} {'F(2x4) fp32':>12s} {'F(4x4) fp32':>12s}")
for (B, C, N, S) in [(4, 64, 64, 64), (4, 128, 128, 32), (8, 256, 256, 16), (8, 512, 512, 8), (16, 512, 512, 4), (8, 1024, 256, 8)]:
    x = torch.randn(B, C, S, S, dtype=f64)
    x = F.leaky_relu(x, 0.2)                                            # activations: post-LeakyReLU
    w = torch.randn(N, C, 3, 3, dtype=f64) / (C * 9) ** 0.5
    ref = F.conv2d(x, w, padding=1)
    d = F.conv2d(x.float(), w.float(), padding=1)
    a22 = wino(x, w, BT2, G2, AT2, BT2, G2, AT2, torch.float32)
    a24 = wino(x, w, BT2, G2, AT2, BT4, G4, AT4, torch.float32)
    a44 = wino(x, w, BT4, G4, AT4, BT4, G4, AT4, torch.float32)
    assert err(wino(x, w, BT2, G2, AT2, BT4, G4, AT4, f64), ref) < 1e-12      # the algebra itself
    print(f"{C:5d} -> {N:4d} @ {S:3d}x{S:<3d}       {err(d, ref):12.2e} {err(a22, ref):12.2e} {err(a24, ref):12.2e} {err(a44, ref):12.2e}")

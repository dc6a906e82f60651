"""Replayable record of the torch CPU random draws made while a module is constructed.

TEST INFRASTRUCTURE ONLY (see oracle/mtdgan_oracle.py's header for who may import oracle/).

The reference initialises its networks from the GLOBAL torch generator in its own module-construction order
(nn.Conv2d's default init, spectral_norm's u / v start vectors, then MTD_GAN's `__init_weights` re-draw).  The
product builds the same 326 tensors in a different order, so `torch.manual_seed(s)` alone does not give it the
reference's default-init state.  oracle/pin_against_reference.py therefore records, while the REFERENCE constructs
its model, every in-place draw as (op, shape, a, b) -- the tape -- and how each final state tensor derives from one of
the draws (identity, L2-normalised, or a constant fill).  `replay()` re-issues the same calls on fresh tensors after
the same seed: same torch build, same generator algorithm, hence the same numbers, with nothing of the reference's
code involved.  The fixture also holds a float64 checksum per tensor so that a torch build whose generator differs
fails loudly on the state, not mysteriously on a loss value."""
import contextlib

import torch
import torch.nn.functional as F


@contextlib.contextmanager
def recording(tape):
    """Log every Tensor.normal_ / Tensor.uniform_ call (the only draws nn.Module constructors and nn.init make)."""
    real_n, real_u = torch.Tensor.normal_, torch.Tensor.uniform_

    def normal_(self, mean=0.0, std=1.0, *, generator=None):
        out = real_n(self, mean, std, generator=generator)
        if generator is None:
            tape.append(["normal_", list(self.shape), float(mean), float(std)])
            tape_values.append(self.detach().clone())
        return out

    def uniform_(self, a=0.0, b=1.0, *, generator=None):
        out = real_u(self, a, b, generator=generator)
        if generator is None:
            tape.append(["uniform_", list(self.shape), float(a), float(b)])
            tape_values.append(self.detach().clone())
        return out
    tape_values = []
    torch.Tensor.normal_, torch.Tensor.uniform_ = normal_, uniform_
    try:
        yield tape_values
    finally:
        torch.Tensor.normal_, torch.Tensor.uniform_ = real_n, real_u


def derive(state, draws):
    """For every tensor of `state`: which draw it is.  Returns {name: [kind, arg, shape]} with kind 'same' (bit-identical
    to draw number arg), 'unit' (F.normalize(draw, dim=0, eps=1e-12): spectral_norm's u / v) or 'fill' (constant arg).
    Later draws win (a re-initialisation overwrites the constructor's default init)."""
    by_shape = {}
    for i, d in enumerate(draws):
        by_shape.setdefault(tuple(d.shape), []).append(i)
    out = {}
    for name, t in state.items():
        hit = None
        for i in reversed(by_shape.get(tuple(t.shape), [])):
            if torch.equal(draws[i], t):
                hit = ["same", i, list(t.shape)]
                break
            if t.dim() == 1 and torch.equal(F.normalize(draws[i], dim=0, eps=1e-12), t):
                hit = ["unit", i, list(t.shape)]
                break
        if hit is None:
            v = t.reshape(-1)[0].item() if t.numel() else 0.0
            if not torch.equal(t, torch.full_like(t, v)):
                raise AssertionError(f"{name}: not one of the recorded draws and not a constant fill")
            hit = ["fill", v, list(t.shape)]
        out[name] = hit
    return out


def replay(tape, derivation, dtype=torch.float32):
    """Re-issue the taped draws from the CURRENT global generator state and assemble the state dict."""
    draws = []
    for op, shape, a, b in tape:
        t = torch.empty(shape, dtype=dtype)
        draws.append(t.normal_(a, b) if op == "normal_" else t.uniform_(a, b))
    state = {}
    for name, (kind, arg, shape) in derivation.items():
        if kind == "same":
            state[name] = draws[arg]
        elif kind == "unit":
            state[name] = F.normalize(draws[arg], dim=0, eps=1e-12)
        else:
            state[name] = torch.full(shape, arg, dtype=dtype)
    return state


def checksum(t):
    d = t.double()
    return [float(d.sum()), float(d.abs().sum())]

"""Error measures shared by the GPU parity tests."""
import torch


def rel(a, b):
    """The larger of two errors of a against the reference b:
      * max |a - b| / max |b|                         (a tensor-wide bound), and
      * max over elements of |a - b| / (|b| + rms(b))  (element-wise: relative for the elements that carry the tensor's
        energy, absolute at the scale of the tensor's RMS -- not of its largest element -- for the small ones, so an error
        hidden under one large entry, or spread over many small ones, still shows)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    d = (a - b).abs()
    e_max = d.max().item() / (b.abs().max().item() + 1e-30)
    rms = b.pow(2).mean().sqrt().item()
    e_el = (d / (b.abs() + rms + 1e-30)).max().item()
    return max(e_max, e_el)

"""Pixel metrics of the reference's evaluation loops -- mirror of metrics.py:172-244 (compute_RMSE / compute_PSNR /
compute_SSIM: same arguments, same (input, gt, pred) float triples) on the HIP kernel `mtd_image_metrics`.
The perceptual / feature metrics (PL, TML, FID: VGG16 / InceptionV3 from torchvision weights) are out of scope."""
import numpy as np
import torch

from . import _lib
from . import kernels as K


def _pair_sums(a, b, clip_a=False):
    """(sum of squared error, sum of the SSIM map, pixel count) of two (B,1,H,W) CUDA tensors; one device->host copy."""
    if a.dim() != 4 or b.dim() != 4 or a.shape != b.shape or a.shape[1] != 1:
        raise AssertionError("pixel metrics expect two (B,1,H,W) tensors of the same shape")
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("pixel metrics: HIP path needs CUDA tensors (no CPU fallback)")
    a, b = a.contiguous().float(), b.contiguous().float()
    B, _, H, W = a.shape
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    ws = K.workspace(L.mtd_image_metrics_ws_bytes(B, H, W), a.device)
    _lib.check(L.mtd_image_metrics(a.data_ptr(), b.data_ptr(), B, H, W, 1 if clip_a else 0, out.data_ptr(), ws.data_ptr(), K.stream_ptr()),
               "mtd_image_metrics")
    sse, ssum = out.tolist()
    return sse, ssum, B * H * W


def _psnr(sse, n, data_range):
    mse = np.float32(np.float32(sse / n) + np.float32(1e-10))
    return float(np.float32(10.0) * np.log10(np.float32(data_range ** 2) / mse))


def compute_RMSE(input, target, pred):
    """metrics.py:174-181."""
    r = []
    for a in (input, target, pred):
        sse, _, n = _pair_sums(a, target)
        r.append(float(np.sqrt(np.float32(sse / n))))
    return tuple(r)


def compute_PSNR(input, target, pred, data_range=1.0):
    """metrics.py:184-197: MSE over the whole batch tensor, + 1e-10."""
    r = []
    for a in (input, target, pred):
        sse, _, n = _pair_sums(a, target)
        r.append(_psnr(sse, n, data_range))
    return tuple(r)


def compute_SSIM(input, target, pred, data_range=1.0):
    """metrics.py:200-244 (window 11, sigma 1.5, size_average)."""
    if data_range != 1.0:
        raise NotImplementedError("SSIM kernel is built for data_range 1.0 (C1 = 1e-4, C2 = 9e-4), as the reference calls it")
    r = []
    for a in (input, target, pred):
        _, ssum, n = _pair_sums(a, target)
        r.append(float(np.float32(ssum / n)))
    return tuple(r)


def pixel_metrics(input, target, pred):
    """All nine numbers of the test loop in three launches: dict of (input, gt, pred) triples."""
    out = {"rmse": [], "psnr": [], "ssim": []}
    for a in (input, target, pred):
        sse, ssum, n = _pair_sums(a, target)
        out["rmse"].append(float(np.sqrt(np.float32(sse / n))))
        out["psnr"].append(_psnr(sse, n, 1.0))
        out["ssim"].append(float(np.float32(ssum / n)))
    return {k: tuple(v) for k, v in out.items()}

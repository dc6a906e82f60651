"""Device-side mirror of the reference's Mayo training front end (create_datasets/Mayo.py:117-136, type "window_patch"):
HU window [-160, 240] -> [0, 1], crop to the full-dose foreground, pad to >= 64, 8 random 64 x 64 samples per slice, then per
sample RandRotate90d / RandFlipd / RandRotated (prob 0.1 each).  The reference builds this from monai 1.3.2 transforms on the
CPU; here one HIP gather (include/mtdgan_hip.h: mtd_foreground_bbox, mtd_window_patches) produces the batch from HU slices
that are already resident in HBM.  DICOM decoding (pydicom, Mayo.py:19-43) stays on the host: get_pixels_hu's int16 HU
arrays are this module's input.  No CPU fallback."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .. import kernels as K

A_MIN, A_MAX = -160.0, 240.0          # Mayo.py:119  ScaleIntensityRanged(a_min=-160, a_max=240, b_min=0, b_max=1, clip=True)
ROI, NUM_SAMPLES, PROB = 64, 8, 0.1   # Mayo.py:126-131
MAX_ANGLE = math.pi / 12              # Mayo.py:131  range_x


def draw_descriptors(rng, n_slices, num_samples=NUM_SAMPLES, prob=PROB):
    """The random choices of one batch, per sample in the pipeline's order (crop origin y, x; quarter turns; flip; angle).
    rng: numpy RandomState / Generator-like with random() and randint()/integers().  (monai draws from its own
    RandomState in a different interleaving; the choices are injectable so that tests and the oracle share them.)"""
    randint = getattr(rng, "integers", None) or rng.randint
    out = []
    for s in range(n_slices):
        for _ in range(num_samples):
            uy, ux = float(rng.random()), float(rng.random())
            k = int(randint(1, 4)) if rng.random() < prob else 0
            flip = 1 if rng.random() < prob else 0
            ang = float(rng.uniform(-MAX_ANGLE, MAX_ANGLE)) if rng.random() < prob else 0.0
            out.append((s, uy, ux, k, flip, ang))
    return out


def window_patches(hu_low, hu_full, descriptors, roi=ROI, a_min=A_MIN, a_max=A_MAX):
    """hu_low / hu_full: (S, H, W) int16 CUDA tensors (quarter / full dose).  descriptors: list of
    (slice, uy, ux, rot_k, flip, angle).  Returns (n_20, n_100): two (len(descriptors), 1, roi, roi) float32 tensors --
    what list_data_collate hands the training loop as batch["n_20"], batch["n_100"] (Mayo.py:176)."""
    if not (hu_low.is_cuda and hu_full.is_cuda and hu_low.dtype == torch.int16 and hu_full.dtype == torch.int16):
        raise RuntimeError("window_patches: HIP path needs int16 CUDA tensors of Hounsfield units")
    if hu_low.shape != hu_full.shape or hu_low.dim() != 3:
        raise ValueError("window_patches: expected two (S, H, W) tensors of the same shape")
    hu_low, hu_full = hu_low.contiguous(), hu_full.contiguous()
    S, H, W = hu_low.shape
    dev = hu_low.device
    L = _lib.lib()
    n = len(descriptors)
    structs = []
    for (s, uy, ux, k, flip, ang) in descriptors:
        if not (0 <= s < S):
            raise ValueError("descriptor refers to a slice outside the batch")
        d = _lib.PatchDesc()
        d.slice, d.uy, d.ux, d.rot_k, d.flip, d.angle = int(s), float(uy), float(ux), int(k), int(flip), float(ang)
        structs.append(d)
    tab, _host = K.device_table(structs, dev)
    bbox = torch.empty((S, 4), dtype=torch.int32, device=dev)
    K.check(L.mtd_foreground_bbox(hu_full.data_ptr(), S, H, W, a_min, bbox.data_ptr(), K.stream_ptr()), "mtd_foreground_bbox")
    lo = torch.empty((n, 1, roi, roi), dtype=torch.float32, device=dev)
    hi = torch.empty((n, 1, roi, roi), dtype=torch.float32, device=dev)
    K.check(L.mtd_window_patches(hu_low.data_ptr(), hu_full.data_ptr(), S, H, W, bbox.data_ptr(), tab.data_ptr(), n, a_min, a_max, roi,
                                 lo.data_ptr(), hi.data_ptr(), K.stream_ptr()), "mtd_window_patches")
    return lo, hi


def window_slices(hu, a_min=A_MIN, a_max=A_MAX):
    """Validation / test pipeline (Mayo.py:150-157): whole slices, window only.  hu: (..., H, W) int16 CUDA."""
    if not (hu.is_cuda and hu.dtype == torch.int16):
        raise RuntimeError("window_slices: HIP path needs an int16 CUDA tensor")
    hu = hu.contiguous()
    out = torch.empty(hu.shape, dtype=torch.float32, device=hu.device)
    K.check(_lib.lib().mtd_hu_window(hu.data_ptr(), hu.numel(), a_min, a_max, out.data_ptr(), K.stream_ptr()), "mtd_hu_window")
    return out


def synthetic_hu_slices(n, size=512, seed=0):
    """LDCT-shaped int16 HU slice pairs for tests and benchmarks (no patient data in this repository): a smooth body
    ellipse in air, quarter-dose = full dose + noise."""
    g = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    lo, hi = [], []
    for i in range(n):
        cy, cx = size * (0.45 + 0.1 * g.rand()), size * (0.45 + 0.1 * g.rand())
        ry, rx = size * (0.25 + 0.1 * g.rand()), size * (0.3 + 0.1 * g.rand())
        body = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        tex = g.randn(size // 8, size // 8).astype(np.float32).repeat(8, 0).repeat(8, 1)
        full = np.where(body, 40.0 + 120.0 * tex, -1000.0)
        low = full + np.where(body, 40.0 * g.randn(size, size), 5.0 * g.randn(size, size))
        hi.append(np.clip(np.round(full), -1024, 3071).astype(np.int16))
        lo.append(np.clip(np.round(low), -1024, 3071).astype(np.int16))
    return torch.from_numpy(np.stack(lo)), torch.from_numpy(np.stack(hi))

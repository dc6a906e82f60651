"""TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline): CPU restatement of the reference's "window_patch" training
front end, create_datasets/Mayo.py:117-136.

PARITY: the transforms live in the third-party dependency monai (requirements.txt: monai==1.3.2), which is not in this
image and cannot be imported to generate golden vectors, and the reference holds no fixtures for this path.  What is restated
is monai's published behaviour, and the one part of it that is an executable torch call is pinned to that call:
  ScaleIntensityRange(a_min, a_max, 0, 1, clip=True):  (x - a_min) / (a_max - a_min), clipped to [0, 1], float32
  CropForeground(select_fn = x > 0, margin 0):         bounding box [min, max + 1) of the foreground of the full-dose key
  SpatialPad(spatial_size, method="symmetric"):        zeros, floor(half) before and the rest after
  RandSpatialCropSamples(roi, random_center=True):     origin uniform in [0, size - roi]
  RandRotate90(prob, max_k=3), axes (0, 1):            np.rot90(img, k)            == torch.rot90 on the spatial axes (test)
  RandFlip(prob, spatial_axis=[0, 1]):                 np.flip over both axes      == torch.flip on the spatial axes (test)
  RandRotate(prob, range_x, keep_size=True, "bilinear", "border", align_corners=False): rotation about the centre
      -- monai builds T = shift((n-1)/2) @ [[cos, -sin], [sin, cos]] @ shift(-(n-1)/2) as the OUTPUT -> INPUT index map on
      (row, col), normalises it (i -> (2 i + 1) / n - 1), reverses the index order and calls torch.nn.functional.affine_grid /
      grid_sample(bilinear, border, align_corners=False).  tests/test_data_front_end.py builds exactly that with torch and
      checks rotate_bilinear_border() against it (direction, centre, border clamp; 1e-5 on values in [0, 1]).
The random draws are inputs (descriptors), shared with the device path; monai's own RandomState interleaving is not
reproduced.  Status: "pinned to torch's affine_grid / grid_sample through monai's published matrix construction; monai itself
absent" -- short of golden vectors from monai, which this image cannot produce."""
import numpy as np


def window(hu, a_min=-160.0, a_max=240.0):
    v = (hu.astype(np.float32) - np.float32(a_min)) / np.float32(a_max - a_min)
    return np.clip(v, np.float32(0), np.float32(1)).astype(np.float32)


def foreground_bbox(hu_full, a_min=-160.0):
    fg = hu_full.astype(np.float32) > np.float32(a_min)
    if not fg.any():
        return 0, hu_full.shape[0], 0, hu_full.shape[1]
    ys, xs = np.where(fg.any(1))[0], np.where(fg.any(0))[0]
    return int(ys[0]), int(ys[-1]) + 1, int(xs[0]), int(xs[-1]) + 1


def window_patches(hu_low, hu_full, descriptors, roi=64, a_min=-160.0, a_max=240.0):
    """hu_low / hu_full: (S, H, W) int16 arrays.  Returns two (n, 1, roi, roi) float32 arrays."""
    outs_lo, outs_hi = [], []
    for (s, uy, ux, k, flip, ang) in descriptors:
        y0, y1, x0, x1 = foreground_bbox(hu_full[s], a_min)
        pair = []
        for img in (hu_low[s], hu_full[s]):
            crop = window(img, a_min, a_max)[y0:y1, x0:x1]
            ph, pw = max(crop.shape[0], roi), max(crop.shape[1], roi)
            py, px = (ph - crop.shape[0]) // 2, (pw - crop.shape[1]) // 2
            pad = np.zeros((ph, pw), np.float32)
            pad[py:py + crop.shape[0], px:px + crop.shape[1]] = crop
            ry, rx = ph - roi + 1, pw - roi + 1
            oy = min(int(np.float32(uy) * np.float32(ry)), ry - 1)
            ox = min(int(np.float32(ux) * np.float32(rx)), rx - 1)
            patch = pad[oy:oy + roi, ox:ox + roi]
            if k:
                patch = np.rot90(patch, k)
            if flip:
                patch = patch[::-1, ::-1]
            if ang != 0.0:
                patch = rotate_bilinear_border(np.ascontiguousarray(patch), ang)
            pair.append(np.ascontiguousarray(patch, dtype=np.float32))
        outs_lo.append(pair[0][None])
        outs_hi.append(pair[1][None])
    return np.stack(outs_lo), np.stack(outs_hi)


def rotate_bilinear_border(img, angle):
    R = img.shape[0]
    c = np.float32(0.5 * (R - 1))
    cs, sn = np.float32(np.cos(np.float32(angle))), np.float32(np.sin(np.float32(angle)))
    yy, xx = np.mgrid[0:R, 0:R].astype(np.float32)
    dy, dx = yy - c, xx - c
    sy = np.clip(c + cs * dy - sn * dx, 0, R - 1).astype(np.float32)
    sx = np.clip(c + sn * dy + cs * dx, 0, R - 1).astype(np.float32)
    iy0, ix0 = np.floor(sy).astype(np.int64), np.floor(sx).astype(np.int64)
    iy1, ix1 = np.minimum(iy0 + 1, R - 1), np.minimum(ix0 + 1, R - 1)
    fy, fx = sy - iy0.astype(np.float32), sx - ix0.astype(np.float32)
    one = np.float32(1)
    return (((img[iy0, ix0] * ((one - fy) * (one - fx)) + img[iy0, ix1] * ((one - fy) * fx)) + img[iy1, ix0] * (fy * (one - fx)))
            + img[iy1, ix1] * (fy * fx)).astype(np.float32)

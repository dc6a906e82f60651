"""Training-patch front end (create_datasets/Mayo.py:117-136 "window_patch"): the CPU restatement on its own (CPU suite) and
the HIP gather against it (GPU suite).  monai is not in this image: parity is against the restated published behaviour."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import data_oracle as DO


def _slices(n=3, size=160, seed=3):
    from mtd_gan_amd.create_datasets import Mayo
    lo, hi = Mayo.synthetic_hu_slices(n, size=size, seed=seed)
    return lo, hi


def test_oracle_window_and_box():
    hu = np.array([[-1000, -160, -159], [40, 240, 3000]], np.int16)
    w = DO.window(hu)
    assert w.dtype == np.float32
    assert w[0, 0] == 0.0 and w[0, 1] == 0.0 and w[1, 1] == 1.0 and w[1, 2] == 1.0
    assert w[0, 2] == np.float32(1.0) / np.float32(400.0) and w[1, 0] == np.float32(200.0) / np.float32(400.0)
    assert DO.foreground_bbox(hu) == (0, 2, 0, 3)                     # -159 is foreground (x > 0 after the window)
    assert DO.foreground_bbox(np.full((4, 5), -500, np.int16)) == (0, 4, 0, 5)


def test_oracle_pipeline_shapes_and_moves():
    lo, hi = _slices(2, size=96)
    lo, hi = lo.numpy(), hi.numpy()
    base = [(0, 0.3, 0.6, 0, 0, 0.0)]
    a_lo, a_hi = DO.window_patches(lo, hi, base)
    assert a_lo.shape == (1, 1, 64, 64) and a_hi.dtype == np.float32
    assert 0.0 <= a_lo.min() and a_hi.max() <= 1.0
    r_lo, _ = DO.window_patches(lo, hi, [(0, 0.3, 0.6, 1, 0, 0.0)])
    assert np.array_equal(r_lo[0, 0], np.rot90(a_lo[0, 0], 1))
    f_lo, _ = DO.window_patches(lo, hi, [(0, 0.3, 0.6, 0, 1, 0.0)])
    assert np.array_equal(f_lo[0, 0], a_lo[0, 0][::-1, ::-1])
    t_lo, _ = DO.window_patches(lo, hi, [(0, 0.3, 0.6, 0, 0, 1e-9)])   # a vanishing angle reproduces the patch
    assert np.abs(t_lo - a_lo).max() < 1e-6
    # a foreground smaller than the sample: symmetric zero padding
    small = np.full((1, 96, 96), -1000, np.int16)
    small[0, 40:50, 30:70] = 100
    p_lo, _ = DO.window_patches(small, small, [(0, 0.0, 0.0, 0, 0, 0.0)])
    assert p_lo[0, 0, 27:37, 12:52].min() > 0 and p_lo[0, 0, :27].max() == 0 and p_lo[0, 0, :, :12].max() == 0


def _monai_rotate_2d(img, angle):
    """monai 1.3.2's 2-D `Rotate(angle, keep_size=True, mode="bilinear", padding_mode="border", align_corners=False)` -- what
    `RandRotated(range_x=pi/12, keep_size=True, align_corners=False)` of create_datasets/Mayo.py:128 applies -- restated down to
    the torch calls monai makes, and executed with torch's own affine_grid / grid_sample:
      transforms/spatial/functional.py rotate():   T = shift(+(n-1)/2) @ create_rotate(2, angle) @ shift(-(n-1)/2), an index-space
                                                   map from OUTPUT (row, col) to INPUT (row, col), create_rotate = [[c, -s], [s, c]];
      networks/layers/spatial_transforms.py AffineTransform(normalized=False, reverse_indices=True):
                                                   to_norm_affine(T, src, dst, align_corners) then rows and columns reversed
                                                   ((row, col) -> torch's (x, y)), affine_grid, grid_sample;
      networks/utils.py normalize_transform(shape, align_corners=False, zero_centered=False):  i -> (2 i + 1) / n - 1."""
    import math
    n = img.shape[-1]
    c, s_ = math.cos(angle), math.sin(angle)
    R = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    sh = lambda v: torch.tensor([[1.0, 0.0, v], [0.0, 1.0, v], [0.0, 0.0, 1.0]], dtype=torch.float64)
    T = sh((n - 1) / 2) @ R @ sh(-(n - 1) / 2)
    norm = torch.tensor([[2.0 / n, 0.0, 1.0 / n - 1.0], [0.0, 2.0 / n, 1.0 / n - 1.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    theta = norm @ T @ torch.linalg.inv(norm)
    theta = theta[[1, 0, 2]][:, [1, 0, 2]]                    # reverse_indices
    t = torch.from_numpy(np.ascontiguousarray(img))[None, None].double()
    grid = torch.nn.functional.affine_grid(theta[None, :2], (1, 1, n, n), align_corners=False)
    out = torch.nn.functional.grid_sample(t, grid, mode="bilinear", padding_mode="border", align_corners=False)
    return out[0, 0].float().numpy()


def test_rotation_convention_matches_affine_grid_and_grid_sample():
    """f3 pin: the restated rotation (direction, centre, border handling) equals torch's affine_grid + grid_sample driven
    the way monai's Rotate drives them; quarter turns and flips equal torch.rot90 / torch.flip on the spatial axes (0, 1)
    of a channel-first image (monai RandRotate90d(spatial_axes=[0, 1]) / RandFlipd(spatial_axis=[0, 1]))."""
    rng = np.random.RandomState(4)
    for n in (64, 17):
        img = rng.rand(n, n).astype(np.float32)
        for ang in (0.26, -0.26, 0.1, -0.013, 0.0):
            want = _monai_rotate_2d(img, ang)
            got = DO.rotate_bilinear_border(img, ang)
            assert np.abs(got - want).max() < 1e-5, (n, ang, np.abs(got - want).max())      # fp32 coordinates vs float64
        # a direction check that does not depend on interpolation: a bright pixel right of the centre moves UP for angle > 0
        dot = np.zeros((65, 65), np.float32)
        dot[32, 52] = 1.0
        up = DO.rotate_bilinear_border(dot, 0.25)
        yy, xx = np.unravel_index(np.argmax(up), up.shape)
        ref = _monai_rotate_2d(dot, 0.25)
        assert (yy, xx) == tuple(np.unravel_index(np.argmax(ref), ref.shape)) and yy != 32
    lo, hi = _slices(1, size=96)
    base = [(0, 0.3, 0.6, 0, 0, 0.0)]
    a, _ = DO.window_patches(lo.numpy(), hi.numpy(), base)
    chw = torch.from_numpy(a[0])                                              # (1, 64, 64) channel first
    for k in (1, 2, 3):
        r, _ = DO.window_patches(lo.numpy(), hi.numpy(), [(0, 0.3, 0.6, k, 0, 0.0)])
        assert np.array_equal(r[0], torch.rot90(chw, k, (1, 2)).numpy())
    f, _ = DO.window_patches(lo.numpy(), hi.numpy(), [(0, 0.3, 0.6, 0, 1, 0.0)])
    assert np.array_equal(f[0], torch.flip(chw, (1, 2)).numpy())


@pytest.mark.gpu
def test_patches_vs_oracle(hip_lib):
    from mtd_gan_amd.create_datasets import Mayo
    lo, hi = _slices(3, size=160, seed=5)
    rng = np.random.RandomState(11)
    descs = Mayo.draw_descriptors(rng, 3, num_samples=8, prob=0.5)
    # make sure every transform and their combinations occur
    descs += [(1, 0.99999, 0.0, 3, 1, 0.2), (2, 0.0, 0.99999, 2, 0, -0.26), (0, 0.5, 0.5, 1, 1, 0.0)]
    g_lo, g_hi = Mayo.window_patches(lo.cuda(), hi.cuda(), descs)
    torch.cuda.synchronize()
    o_lo, o_hi = DO.window_patches(lo.numpy(), hi.numpy(), descs)
    for i, d in enumerate(descs):
        a, b = g_lo[i].cpu().numpy(), g_hi[i].cpu().numpy()
        if d[5] == 0.0:                                   # window, box, pad, sample, quarter turns, flip: byte work, exact
            assert np.array_equal(a, o_lo[i]) and np.array_equal(b, o_hi[i]), d
        else:                                             # bilinear rotation: fp32, tolerance 1e-5 absolute (values in [0, 1])
            assert np.abs(a - o_lo[i]).max() < 1e-5 and np.abs(b - o_hi[i]).max() < 1e-5, d


@pytest.mark.gpu
def test_patches_small_foreground_and_empty_slice(hip_lib):
    from mtd_gan_amd.create_datasets import Mayo
    hu = np.full((2, 128, 128), -1000, np.int16)
    hu[0, 50:70, 20:110] = 60                             # 20 x 90 foreground: padded to 64 rows
    lo = torch.from_numpy(hu.copy())
    hi = torch.from_numpy(hu)
    descs = [(0, 0.0, 0.0, 0, 0, 0.0), (0, 0.7, 0.9999, 1, 1, 0.0), (1, 0.4, 0.2, 0, 0, 0.0), (1, 0.4, 0.2, 2, 0, 0.1)]
    g_lo, g_hi = Mayo.window_patches(lo.cuda(), hi.cuda(), descs)
    o_lo, o_hi = DO.window_patches(lo.numpy(), hi.numpy(), descs)
    assert np.array_equal(g_lo.cpu().numpy()[:3], o_lo[:3]) and np.array_equal(g_hi.cpu().numpy()[:3], o_hi[:3])
    assert np.abs(g_lo.cpu().numpy()[3] - o_lo[3]).max() < 1e-5


@pytest.mark.gpu
def test_window_slices_vs_oracle(hip_lib):
    from mtd_gan_amd.create_datasets import Mayo
    lo, _ = _slices(2, size=96, seed=9)
    out = Mayo.window_slices(lo.cuda())
    assert np.array_equal(out.cpu().numpy(), DO.window(lo.numpy()))

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

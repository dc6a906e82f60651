"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch CPU fp32
references of the same op (conv2d / conv_transpose2d / autograd / interpolate / pixel_shuffle) and the
oracle's spectral path.  Tolerance: 1e-3 relative to the tensor's max-abs (north_star), most land ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-3


def nhwc(t):   # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):   # NHWC cuda -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


from _metrics import rel as relerr  # noqa: E402  (tensor-wide AND element-wise bound)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # B, Cin, Cout, H, W, k, s, p
    (2, 32, 32, 64, 64, 3, 1, 1),     # generator conv, cfg 256x32
    (1, 32, 32, 8, 8, 3, 1, 1),       # small-M 128x32, split-K
    (32, 32, 32, 64, 64, 3, 1, 1),    # full batch: 256 weight-gradient slabs, 1024-thread fused reduce
    (2, 64, 64, 32, 32, 3, 1, 1),     # 256x64
    (2, 64, 64, 16, 16, 4, 2, 1),     # strided 4x4, 64x64 tile
    (2, 64, 128, 16, 16, 3, 1, 1),    # 128x128
    (3, 256, 512, 4, 4, 3, 1, 1),     # deep layer, M=48 -> 32x128 tiles, split-K, tap skipping
    (2, 512, 512, 2, 2, 3, 1, 1),
    (2, 512, 512, 2, 2, 4, 2, 1),     # -> 1x1
    (3, 64, 64, 16, 16, 4, 2, 1),     # all-taps weight gradient: 4 tiles, six pixel splits (slabs)
    (2, 512, 512, 4, 4, 4, 2, 1),     # all-taps: 8 pixels, single split, LDS-transposed [n][c][16] rows
    (5, 256, 512, 8, 8, 4, 2, 1),     # all-taps: 80 pixels, two ragged splits
    (2, 512, 512, 1, 1, 1, 1, 0),     # bconv (1x1)
    (2, 128, 256, 8, 8, 1, 1, 0),
    (2, 1, 32, 64, 64, 3, 1, 1),      # direct: Cin = 1
    (2, 128, 1, 16, 16, 3, 1, 1),     # direct: Cout = 1
    (2, 1, 1, 16, 16, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(hip_lib, case):
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, W, k, s, p = case
    x = rnd(B, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5).requires_grad_(True)
    b = rnd(Co, seed=3, scale=0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    ref = F.leaky_relu(y * 0.7, 0.2)     # exercises scale + bias + act (bias is added after the scale)
    # library computes act(scale*conv + bias): fold so both agree
    ref = F.leaky_relu(F.conv2d(x, w, None, stride=s, padding=p) * 0.7 + b.view(1, -1, 1, 1), 0.2)
    cot = rnd(*ref.shape, seed=4)
    (ref * cot).sum().backward()
    OH, OW = y.shape[2:]
    xd, wd, bd = nhwc(x.detach()), w.detach().cuda(), b.detach().cuda()
    scale = torch.tensor([0.7], device="cuda")
    out = torch.empty(B, OH, OW, Co, device="cuda")
    K.conv(xd, wd, K.geom_fwd(B, H, W, k, s, p), Co, Ci, Ci * k * k, k * k, out, scale=scale, bias=bd, act=K.ACT_LRELU)
    assert relerr(nchw(out), ref.detach()) < TOL
    # gradient w.r.t. pre-activation, then dgrad / wgrad
    gpre = K.act_grad(nhwc(cot), out, 0.2)
    gpre_ref = cot * torch.where(ref.detach() > 0, 1.0, 0.2)
    assert relerr(nchw(gpre), gpre_ref) < TOL
    dx = torch.empty(B, H, W, Ci, device="cuda")
    if s == 1:
        K.conv(gpre, wd, K.geom_dgrad_s1(B, H, W, k, p), Ci, Co, k * k, Ci * k * k, dx, scale=scale)
    else:
        for py in range(2):
            for px in range(2):
                K.conv(gpre, wd, K.geom_dgrad_s2(B, H, W, py, px), Ci, Co, k * k, Ci * k * k, dx, scale=scale)
    assert relerr(nchw(dx), x.grad) < TOL
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    K.wgrad(gpre, xd, K.geom_fwd(B, H, W, k, s, p), Co, Ci, dw, Ci * k * k, k * k, db=db)
    assert relerr(dw.cpu() * 0.7, w.grad) < TOL
    assert relerr(db.cpu(), b.grad) < TOL
    # accumulate mode
    K.wgrad(gpre, xd, K.geom_fwd(B, H, W, k, s, p), Co, Ci, dw, Ci * k * k, k * k, db=db, accumulate=True)
    assert relerr(dw.cpu() * 0.35, w.grad) < TOL


@pytest.mark.parametrize("case", [(3, 64, 64, 16, 16, 0), (3, 64, 64, 16, 16, 3), (2, 512, 512, 4, 4, 0), (5, 256, 512, 8, 8, 2), (1, 128, 128, 2, 2, 0),
                                  (3, 64, 96, 32, 32, 0, 15), (3, 32, 64, 32, 32, 5, 15), (2, 128, 64, 16, 16, 1, 15), (1, 64, 64, 64, 64, 7, 15)])
def test_all_taps_weight_gradient_forced(hip_lib, case):
    """wgrad_taps_kernel (one workgroup per (n, c) tile for all 16 taps of a 4x4 stride-2 layer) and wgrad_s2_kernel (the same
    ownership, operands from a double-buffered LDS window) forced through the tuning hook on shapes the plan gives to other
    kernels too: its own pixel split and forced ones (slabs + finish kernels), ragged
    last chunks, a single pixel (2x2 -> 1x1), bias gradient, accumulate mode."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    L = _lib.lib()
    L.mtd_conv_wgrad_override.argtypes = [C.c_int, C.c_int]
    B, Ci, Co, H, W, split = case[:6]
    cfg = case[6] if len(case) > 6 else 13          # 15: the halo-window kernel (8 x 8 pixel blocks from LDS)
    x = rnd(B, Ci, H, W, seed=41)
    w = rnd(Co, Ci, 4, 4, seed=42, scale=0.02).requires_grad_(True)
    b = rnd(Co, seed=43, scale=0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=2, padding=1)
    cot = rnd(*y.shape, seed=44)
    (y * cot).sum().backward()
    try:
        L.mtd_conv_wgrad_override(cfg, split if split else -1)
        dw = torch.full((Co, Ci, 4, 4), 0.25, device="cuda")
        db = torch.full((Co,), 0.5, device="cuda")
        geom = K.geom_fwd(B, H, W, 4, 2, 1)
        K.wgrad(nhwc(cot), nhwc(x), geom, Co, Ci, dw, Ci * 16, 16, db=db)
        assert relerr(dw.cpu(), w.grad) < TOL
        assert relerr(db.cpu(), b.grad) < TOL
        K.wgrad(nhwc(cot), nhwc(x), geom, Co, Ci, dw, Ci * 16, 16, db=db, accumulate=True)
        assert relerr(dw.cpu() * 0.5, w.grad) < TOL
        assert relerr(db.cpu() * 0.5, b.grad) < TOL
    finally:
        L.mtd_conv_wgrad_override(-1, -1)


def test_conv1x1_pixel_shuffle_in_place(hip_lib):
    """conv1x1 (C -> 4C') + PixelShuffle(2) (networks.py:166-175) as four strided-output classes in one grid, written
    into the first C' channels of a wider (concatenated) buffer, against F.pixel_shuffle(F.conv2d)."""
    from mtd_gan_amd import kernels as K
    for (B, Ci, Cu, r) in [(3, 64, 32, 4), (2, 512, 128, 1), (2, 128, 64, 8)]:
        x = rnd(B, Ci, r, r, seed=51)
        w = rnd(4 * Cu, Ci, 1, 1, seed=52, scale=Ci ** -0.5)
        b = rnd(4 * Cu, seed=53)
        ref = F.pixel_shuffle(F.conv2d(x, w, b), 2)
        cat = torch.full((B, 2 * r, 2 * r, Cu + 32), 7.0, device="cuda")
        wd, bd = w.cuda(), b.cuda()
        bu = K.regrouped_bias(bd, 4)
        wq = wd.view(Cu, 4, Ci)
        K.conv_multi([((nhwc(x), wq[:, q], K.geom_pixel_shuffle2(B, r, r, q >> 1, q & 1), Cu, Ci, 4 * Ci, 1, cat[..., :Cu]),
                       dict(bias=bu[q * Cu:(q + 1) * Cu])) for q in range(4)])
        assert relerr(nchw(cat[..., :Cu]), ref) < TOL, (B, Ci, Cu, r)
        assert bool((cat[..., Cu:] == 7.0).all())


def test_strided_wgrad_into_transposed_view(hip_lib):
    """The all-taps weight gradient of a 4x4 stride-2 layer writing an IOHW view (ConvTranspose2d-style parameter: channel
    stride != 16, so the single-split launch scatters value by value instead of copying packed rows), with accumulation."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H = 2, 128, 256, 4
    x = rnd(B, Ci, H, H, seed=31)
    w = rnd(Co, Ci, 4, 4, seed=32, scale=0.02).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=2, padding=1)
    cot = rnd(*y.shape, seed=33)
    (y * cot).sum().backward()
    dwT = torch.full((Ci, Co, 4, 4), 0.5, device="cuda")            # [c][n][ky][kx]: W(n, c, tap) at c * Co * 16 + n * 16 + tap
    K.wgrad(nhwc(cot), nhwc(x), K.geom_fwd(B, H, H, 4, 2, 1), Co, Ci, dwT, 16, Co * 16, accumulate=True)
    assert relerr(dwT.cpu().permute(1, 0, 2, 3) - 0.5, w.grad) < TOL


def test_conv_epilogue_adds_and_mask(hip_lib):
    from mtd_gan_amd import kernels as K
    B, C, H, W = 2, 32, 16, 16
    x, w = rnd(B, C, H, W, seed=5), rnd(C, C, 3, 3, seed=6, scale=0.06)
    a1, a2, m = rnd(B, C, H, W, seed=7), rnd(B, C, H, W, seed=8), rnd(B, C, H, W, seed=9)
    ref = (F.conv2d(x, w, padding=1) + a1 + a2) * torch.where(m > 0, 1.0, 0.2)
    out = torch.empty(B, H, W, C, device="cuda")
    K.conv(nhwc(x), w.cuda(), K.geom_fwd(B, H, W, 3, 1, 1), C, C, C * 9, 9, out, add1=nhwc(a1), add2=nhwc(a2), mask=nhwc(m), mask_slope=0.2)
    assert relerr(nchw(out), ref) < TOL
    # channel-slice views (concat buffers): input read from, and output written into, wider tensors
    wide_in = torch.zeros(B, H, W, 96, device="cuda")
    wide_in[..., 32:64] = nhwc(x)
    wide_out = torch.zeros(B, H, W, 64, device="cuda")
    K.conv(wide_in[..., 32:64], w.cuda(), K.geom_fwd(B, H, W, 3, 1, 1), C, C, C * 9, 9, wide_out[..., 32:64])
    assert relerr(nchw(wide_out[..., 32:64]), F.conv2d(x, w, padding=1)) < TOL
    assert wide_out[..., :32].abs().max().item() == 0.0


def test_conv_transpose_views(hip_lib):
    """ConvTranspose2d(k3,s1,p1) forward / dgrad / wgrad through the strided IOHW weight view."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, W = 2, 32, 32, 16, 16
    for Co in (32, 1):
        x = rnd(B, Ci, H, W, seed=11).requires_grad_(True)
        wt = rnd(Ci, Co, 3, 3, seed=12, scale=0.06).requires_grad_(True)
        ref = F.conv_transpose2d(x, wt, padding=1)
        cot = rnd(*ref.shape, seed=13)
        (ref * cot).sum().backward()
        out = torch.empty(B, H, W, Co, device="cuda")
        gt, gf = K.geom_dgrad_s1(B, H, W, 3, 1), K.geom_fwd(B, H, W, 3, 1, 1)
        K.conv(nhwc(x.detach()), wt.detach().cuda(), gt, Co, Ci, 9, Co * 9, out)
        assert relerr(nchw(out), ref.detach()) < TOL
        dx = torch.empty(B, H, W, Ci, device="cuda")
        K.conv(nhwc(cot), wt.detach().cuda(), gf, Ci, Co, Co * 9, 9, dx)
        assert relerr(nchw(dx), x.grad) < TOL
        dw = torch.empty(Ci, Co, 3, 3, device="cuda")
        K.wgrad(nhwc(cot), nhwc(x.detach()), gt, Co, Ci, dw, 9, Co * 9)
        assert relerr(dw.cpu(), wt.grad) < TOL


def test_linear_as_1x1(hip_lib):
    from mtd_gan_amd import kernels as K
    B = 32
    x = rnd(B, 512, seed=21).requires_grad_(True)
    w = rnd(512, 512, seed=22, scale=0.04).requires_grad_(True)
    b = rnd(512, seed=23, scale=0.1)
    ref = F.linear(x, w, b)
    cot = rnd(B, 512, seed=24)
    (ref * cot).sum().backward()
    xd = x.detach().cuda().reshape(B, 1, 1, 512)
    out = torch.empty(B, 1, 1, 512, device="cuda")
    g = K.geom_fwd(B, 1, 1, 1, 1, 0)
    K.conv(xd, w.detach().cuda(), g, 512, 512, 512, 1, out, bias=b.cuda())
    assert relerr(out.cpu().reshape(B, 512), ref.detach()) < TOL
    dx = torch.empty(B, 1, 1, 512, device="cuda")
    K.conv(cot.cuda().reshape(B, 1, 1, 512), w.detach().cuda(), g, 512, 512, 1, 512, dx)
    assert relerr(dx.cpu().reshape(B, 512), x.grad) < TOL
    dw = torch.empty(512, 512, device="cuda")
    K.wgrad(cot.cuda().reshape(B, 1, 1, 512), xd, g, 512, 512, dw, 512, 1)
    assert relerr(dw.cpu(), w.grad) < TOL
    # Linear 512 -> 1 (enc_out): direct path
    w1 = rnd(1, 512, seed=25, scale=0.04).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    r1 = F.linear(x2, w1)
    c1 = rnd(B, 1, seed=26)
    (r1 * c1).sum().backward()
    o1 = torch.empty(B, 1, 1, 1, device="cuda")
    K.conv(xd, w1.detach().cuda(), g, 1, 512, 512, 1, o1)
    assert relerr(o1.cpu().reshape(B, 1), r1.detach()) < TOL
    dw1 = torch.empty(1, 512, device="cuda")
    db1 = torch.empty(1, device="cuda")
    K.wgrad(c1.cuda().reshape(B, 1, 1, 1), xd, g, 1, 512, dw1, 512, 1, db=db1)
    assert relerr(dw1.cpu(), w1.grad) < TOL
    assert relerr(db1.cpu(), c1.sum().reshape(1)) < TOL
    dx1 = torch.empty(B, 1, 1, 512, device="cuda")
    K.conv(c1.cuda().reshape(B, 1, 1, 1), w1.detach().cuda(), g, 512, 1, 1, 512, dx1)
    assert relerr(dx1.cpu().reshape(B, 512), x2.grad) < TOL


def test_upsample_and_pixel_shuffle(hip_lib):
    from mtd_gan_amd import kernels as K
    for (B, C, H, W) in [(2, 64, 1, 1), (2, 32, 2, 2), (2, 16, 8, 8), (1, 8, 5, 3), (1, 6, 3, 4)]:     # (6 channels: the scalar kernels)
        x = rnd(B, C, H, W, seed=31).requires_grad_(True)
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        cot = rnd(*ref.shape, seed=32)
        (ref * cot).sum().backward()
        out = torch.empty(B, 2 * H, 2 * W, C, device="cuda")
        K.upsample2x_fwd(nhwc(x.detach()), out)
        assert relerr(nchw(out), ref.detach()) < 1e-6
        gin = torch.empty(B, H, W, C, device="cuda")
        K.upsample2x_bwd(nhwc(cot), gin)
        assert relerr(nchw(gin), x.grad) < 1e-6
    x = rnd(2, 64, 4, 4, seed=33)
    ref = F.pixel_shuffle(x, 2)
    out = torch.empty(2, 8, 8, 16, device="cuda")
    K.pixel_shuffle2_fwd(nhwc(x), out)
    assert torch.equal(nchw(out), ref)
    back = torch.empty(2, 4, 4, 64, device="cuda")
    K.pixel_shuffle2_bwd(out, back)
    assert torch.equal(nchw(back), x)
    # the value-by-value kernels (source rows that are not 16-byte aligned)
    wide = torch.zeros(2, 4, 4, 65, device="cuda")
    wide[..., :64] = nhwc(x)
    out2 = torch.empty(2, 8, 8, 16, device="cuda")
    K.pixel_shuffle2_fwd(wide[..., :64], out2)
    assert torch.equal(nchw(out2), ref)
    back2 = torch.zeros(2, 4, 4, 65, device="cuda")
    K.pixel_shuffle2_bwd(out2, back2[..., :64])
    assert torch.equal(nchw(back2[..., :64]), x)
    # strided copy into a concat buffer
    cat = torch.zeros(2, 8, 8, 48, device="cuda")
    K.copy_channels(out, cat[..., 32:48])
    assert torch.equal(cat[..., 32:48].cpu(), out.cpu())
    K.copy_channels(out, cat[..., 32:48], accumulate=True)
    assert torch.equal(cat[..., 32:48].cpu(), 2 * out.cpu())


@pytest.mark.parametrize("four_wave", [True, False])
def test_spectral_path_kernels(hip_lib, four_wave):
    """rows -> columns+mix -> rows against torch.fft on the CPU (forward and backward): the four-wave column / mix kernels
    (csrc/resfft4.hip: quad-split transforms, sign mask) and the one-wave forms; B = 3 so that the last, single-column unit of a
    patch is followed by another patch's data."""
    from mtd_gan_amd import kernels as K
    K.SPECMIX4 = four_wave
    try:
        _spectral_path_case(K, 3 if four_wave else 2, four_wave)
    finally:
        K.SPECMIX4 = True


def _spectral_path_case(K, B, four_wave):
    x = rnd(B, 32, 64, 64, seed=41, scale=0.5).requires_grad_(True)
    w2 = rnd(64, 64, 1, 1, seed=42, scale=0.125).requires_grad_(True)
    b2 = rnd(64, seed=43, scale=0.1).requires_grad_(True)
    f = torch.fft.rfft2(x, s=(64, 64), dim=(2, 3), norm="ortho")
    cat = torch.cat([f.real, f.imag], dim=1)
    z = F.conv2d(cat, w2, b2)
    zr, zi = torch.chunk(F.relu(z), 2, dim=1)
    y = torch.fft.irfft2(torch.complex(zr, zi), s=(64, 64), dim=(2, 3), norm="ortho")
    cot = rnd(*y.shape, seed=44)
    (y * cot).sum().backward()
    xd = nhwc(x.detach())
    R = K.rfft_rows(xd, 0)
    # rows check: R[b, kw, h, 0/1, c] = rfft along W / 8
    fr = torch.fft.rfft(x.detach(), dim=3, norm="ortho")          # (B, C, H, 33)
    Rr = R.cpu().reshape(B, 33, 64, 2, 32)
    assert relerr(Rr[:, :, :, 0, :], fr.real.permute(0, 3, 2, 1)) < 1e-5
    assert relerr(Rr[:, :, :, 1, :], fr.imag.permute(0, 3, 2, 1)) < 1e-5
    w2d = w2.detach().cuda().reshape(64, 64)
    T, S, Z = K.spec_mix_fwd(R, K.transpose64(w2d), b2.detach().cuda(), True)
    Sr = S.cpu().reshape(B, 33, 64, 64)                           # [b, kw, kh, cat]
    assert relerr(Sr, cat.detach().permute(0, 3, 2, 1)) < 1e-5
    if four_wave:          # the saved pre-activation is its sign mask: 128 words per (patch, kw pair), accumulator-block order
        assert Z.dtype == torch.uint8 and Z.numel() == B * 17 * 128 * 8
        words = Z.cpu().numpy().view("<u8").reshape(B, 17, 2, 2, 2, 16)             # [b][pair][col][kh half][out half][register]
        zs = (z.detach().permute(0, 3, 2, 1) > 0).numpy()                            # [b][kw][kh][o]
        for (b_, kw_, kh_, o_) in [(0, 0, 0, 0), (1, 5, 37, 40), (2, 32, 63, 63), (0, 17, 12, 31), (2, 31, 45, 2)]:
            r32 = kh_ & 31
            e_ = (r32 & 3) + 4 * (r32 >> 3)
            bit = (o_ & 31) + 32 * ((r32 >> 2) & 1)
            got = (int(words[b_, kw_ // 2, kw_ % 2, kh_ >> 5, o_ >> 5, e_]) >> bit) & 1
            assert got == int(zs[b_, kw_, kh_, o_]), (b_, kw_, kh_, o_)
    else:
        assert relerr(Z.cpu().reshape(B, 33, 64, 64), z.detach().permute(0, 3, 2, 1)) < 1e-5
    out = torch.empty(B, 64, 64, 32, device="cuda")
    K.irfft_rows(T, out)
    assert relerr(nchw(out), y.detach()) < 1e-5
    # backward
    gR = K.rfft_rows(nhwc(cot), 1)
    dw2 = torch.empty(64, 64, device="cuda")
    db2 = torch.empty(64, device="cuda")
    gT = K.spec_mix_bwd(gR, w2d, S, Z, dw2, db2)
    gx = torch.empty(B, 64, 64, 32, device="cuda")
    K.irfft_rows(gT, gx)
    assert relerr(nchw(gx), x.grad) < 1e-4
    assert relerr(dw2.cpu(), w2.grad.reshape(64, 64)) < 1e-4
    assert relerr(db2.cpu(), b2.grad) < 1e-4


import contextlib


@contextlib.contextmanager
def plain_layer_fused(hip_lib, plan):
    """Since plan 19 (the Winograd 32 x 32 weight gradient, csrc/conv_wgrad_wino32.h) a generator layer WITHOUT a spectral tail goes
    as two launches (kernels.conv_wgrad_fused: `unfuse`); the fused launch stays the product path of the 21 layers that close a
    Res-FFT-Conv block, and its weight-gradient role is the row-window kernel's arithmetic.  Tests of the fused launch on a plain
    layer pin that plan (7: forward taps, 8: mirrored) -- and drop the routing decisions cached under the override afterwards."""
    from mtd_gan_amd import kernels as K
    hip_lib.mtd_conv_wgrad_override(plan, -1)
    try:
        yield
    finally:
        hip_lib.mtd_conv_wgrad_override(-1, -1)
        for k in [k for k in K._igemm_ws_cache if isinstance(k, tuple) and k and k[0] == "c32 unfuse"]:
            del K._igemm_ws_cache[k]



@pytest.mark.gpu
@pytest.mark.parametrize("cfg", list(range(9)))
def test_igemm_every_tile_config(hip_lib, cfg):
    """Each implicit-GEMM instantiation (forced through the tuning hook), with and without split-K, against F.conv2d:
    forward 3x3 / 1x1 / strided 4x4-parity data gradient, on maps with borders inside every tile."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    L = _lib.lib()
    L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
    cases = [(2, 64, 128, 16, 16, 3, 1, 1), (3, 128, 128, 4, 4, 3, 1, 1), (2, 32, 128, 64, 64, 3, 1, 1), (2, 256, 128, 1, 1, 3, 1, 1),
             (2, 64, 128, 8, 8, 1, 1, 0)]
    try:
        for split in (1, 2):
            for (B, Ci, Co, H, W, k, s, p) in cases:
                x = rnd(B, Ci, H, W, seed=11)
                w = rnd(Co, Ci, k, k, seed=12, scale=(Ci * k * k) ** -0.5)
                ref = F.conv2d(x, w, None, stride=s, padding=p)
                out = torch.empty(B, ref.shape[2], ref.shape[3], Co, device="cuda")
                L.mtd_conv_igemm_override(cfg, split)
                K._igemm_ws_cache.clear()
                K.conv(nhwc(x), w.cuda(), K.geom_fwd(B, H, W, k, s, p), Co, Ci, Ci * k * k, k * k, out)
                assert relerr(nchw(out), ref) < TOL, (cfg, split, B, Ci, Co, H, W, k)
            # strided data gradient (2x2 taps per output parity)
            B, Ci, Co, H, W = 2, 64, 128, 16, 16
            x = rnd(B, Ci, H, W, seed=13).requires_grad_(True)
            w = rnd(Co, Ci, 4, 4, seed=14, scale=0.03)
            y = F.conv2d(x, w, None, stride=2, padding=1)
            cot = rnd(*y.shape, seed=15)
            (y * cot).sum().backward()
            dx = torch.empty(B, H, W, Ci, device="cuda")
            for py in range(2):
                for px in range(2):
                    K.conv(nhwc(cot), w.cuda(), K.geom_dgrad_s2(B, H, W, py, px), Ci, Co, 16, Ci * 16, dx)
            assert relerr(nchw(dx), x.grad) < TOL, (cfg, split, "dgrad_s2")
            # the same four parity classes as ONE grid (mtd_conv_igemm_multi), full epilogue: adds, LeakyReLU mask, 1/sigma pair
            if cfg <= 5:
                a1, a2, mk = rnd(B, Ci, H, W, seed=16), rnd(B, Ci, H, W, seed=17), rnd(B, Ci, H, W, seed=18)
                sc = torch.tensor([0.7, 1.3], device="cuda")
                dx2 = torch.empty(B, H, W, Ci, device="cuda")
                calls = []
                for py in range(2):
                    for px in range(2):
                        gp = K.geom_dgrad_s2(B, H, W, py, px)
                        calls.append(((nhwc(cot), w.cuda(), gp, Ci, Co, 16, Ci * 16, dx2),
                                      dict(add1=nhwc(a1), add2=nhwc(a2), mask=nhwc(mk), mask_slope=0.2, scale=sc[0:1], scale2=sc[1:2],
                                           scale_split=(B // 2) * gp.OH * gp.OW)))
                K.conv_multi(calls)
                half = torch.cat([torch.full((B // 2, 1, 1, 1), 0.7), torch.full((B - B // 2, 1, 1, 1), 1.3)])
                want = (x.grad * half + a1 + a2) * torch.where(mk > 0, 1.0, 0.2)
                assert relerr(nchw(dx2), want) < TOL, (cfg, split, "dgrad_s2 multi")
    finally:
        L.mtd_conv_igemm_override(-1, -1)
        K._igemm_ws_cache.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 8])
def test_splitk_finish_in_kernel_same_bits(hip_lib, cfg):
    from mtd_gan_amd import _lib as _L
    if not _L.lib().mtd_lab_build():
        pytest.skip("the in-kernel split-K finish of the implicit GEMM exists in lab builds only since round 5 (it lost twice)")
    """Split-K launches that finish inside the kernel (arrival counters, mtd_conv_args.tile_ctr: the last slice to reach a
    tile sums the slabs in slice order and runs the epilogue) against the separate epilogue launch: the same bits, with
    the whole epilogue (1/sigma pair, bias, adds, LeakyReLU mask), ragged last tiles, unaligned output views (scalar form),
    and the counters back at zero afterwards."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    L = _lib.lib()
    L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
    cases = [(3, 128, 128, 4, 4, 3, 1, 1, 2), (2, 512, 128, 2, 2, 3, 1, 1, 8), (5, 256, 64, 3, 3, 3, 1, 1, 4), (2, 64, 128, 8, 8, 1, 1, 0, 2)]
    ctr = K.tile_counters(torch.device("cuda", 0))
    fin_default = K.SPLITK_FIN
    try:
        for (B, Ci, Co, H, W, k, s, p, split) in cases:
            x = rnd(B, Ci, H, W, seed=21)
            w = rnd(Co, Ci, k, k, seed=22, scale=(Ci * k * k) ** -0.5)
            a1, a2, mk = (nhwc(rnd(B, Co, H, W, seed=23 + i)) for i in range(3))
            bias = rnd(Co, seed=26).cuda()
            sc = torch.tensor([0.7, 1.3], device="cuda")
            geom = K.geom_fwd(B, H, W, k, s, p)
            kw = dict(add1=a1, add2=a2, mask=mk, mask_slope=0.2, bias=bias, act=0, scale=sc[0:1], scale2=sc[1:2], scale_split=(B // 2) * H * W)
            L.mtd_conv_igemm_override(cfg, split)
            for unaligned in (False, True):
                outs = []
                for fin in (False, True):
                    K._igemm_ws_cache.clear()
                    K.SPLITK_FIN = fin
                    big = torch.zeros(B, H, W, Co + 4, device="cuda")
                    out = big[..., 1:Co + 1] if unaligned else big[..., :Co]
                    K.conv(nhwc(x), w.cuda(), geom, Co, Ci, Ci * k * k, k * k, out, **kw)
                    outs.append(out.clone())
                assert torch.equal(outs[0], outs[1]), (cfg, split, B, Ci, Co, H, unaligned)
                ref = F.conv2d(x, w, None, stride=s, padding=p)
                half = torch.cat([torch.full((B // 2, 1, 1, 1), 0.7), torch.full((B - B // 2, 1, 1, 1), 1.3)])
                want = (ref * half + bias.cpu().reshape(1, -1, 1, 1) + nchw(a1) + nchw(a2)) * torch.where(nchw(mk) > 0, 1.0, 0.2)
                assert relerr(nchw(outs[1]), want) < TOL, (cfg, split, B, Ci, Co, H, unaligned)
            assert int(ctr.abs().sum()) == 0, "arrival counters must be left at zero"
    finally:
        K.SPLITK_FIN = fin_default
        L.mtd_conv_igemm_override(-1, -1)
        K._igemm_ws_cache.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [-1, 9, 10])
def test_igemm_generator_shaped_kernels(hip_lib, cfg):
    """The two kernels for the generator's 32-channel 3x3 layers on 64 x 64 maps (9 = persistent, 10 = halo tile, -1 = the
    library's choice): forward with the full epilogue, 64 output channels, and the stride-1 data gradient, against F.conv2d."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    L = _lib.lib()
    L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
    B, H = 20, 64            # 320 four-row tiles: the halo-tile kernel's workgroups take one or two each
    try:
        L.mtd_conv_igemm_override(cfg, 1)
        K._igemm_ws_cache.clear()
        for Co in (32, 64):
            x = rnd(B, 32, H, H, seed=21)
            w = rnd(Co, 32, 3, 3, seed=22, scale=0.06)
            bias = rnd(Co, seed=23)
            res = rnd(B, Co, H, H, seed=24)
            ref = F.leaky_relu(F.conv2d(x, w, bias, padding=1) + res, 0.2)
            out = torch.empty(B, H, H, Co, device="cuda")
            K.conv(nhwc(x), w.cuda(), K.geom_fwd(B, H, H, 3, 1, 1), Co, 32, 32 * 9, 9, out, bias=bias.cuda(), add1=nhwc(res),
                   act=_lib.ACT_LRELU)
            assert relerr(nchw(out), ref) < TOL, (cfg, Co)
        # data gradient of a 32 -> 32 layer (flipped taps), input a 32-channel slice of a wider tensor
        x = rnd(B, 32, H, H, seed=25).requires_grad_(True)
        w = rnd(32, 32, 3, 3, seed=26, scale=0.06)
        y = F.conv2d(x, w, None, padding=1)
        cot = rnd(*y.shape, seed=27)
        (y * cot).sum().backward()
        wide = torch.zeros(B, H, H, 48, device="cuda")
        wide[..., 8:40] = nhwc(cot)
        dx = torch.empty(B, H, H, 32, device="cuda")
        K.conv(wide[..., 8:40], w.cuda(), K.geom_dgrad_s1(B, H, H, 3, 1), 32, 32, 9, 32 * 9, dx)
        assert relerr(nchw(dx), x.grad) < TOL, (cfg, "dgrad")
    finally:
        L.mtd_conv_igemm_override(-1, -1)
        K._igemm_ws_cache.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 20, 32])
def test_fused_c32_backward_launch(hip_lib, B):
    """csrc/conv_c32_bwd.hip: the data gradient and the weight + bias gradient of a 32 -> 32 channel 3x3 generator layer in
    ONE launch against the two launches it replaces -- both layer kinds (Conv2d: forward geometry for the weight gradient,
    ConvTranspose2d: mirrored taps), full data-gradient epilogue (add, mask, second output).  B = 20: 320 tiles on 256
    workgroups (workgroups with one and with two live tiles); B = 32: the stand-alone launch's own split, bit for bit."""
    from mtd_gan_amd import kernels as K
    H = W = 64
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen).cuda()
    p_, q_, add1, mask = r(B, H, W, 32), r(B, H, W, 32), r(B, H, W, 32), r(B, H, W, 32)
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.1).cuda()
    gf, gt = K.geom_fwd(B, H, W, 3, 1, 1), K.geom_dgrad_s1(B, H, W, 3, 1)
    for kind in ("conv", "convT"):
        second = B * H * W >= 32768          # (the unfused halo-tile launch takes a second output from 32768 pixels on)
        if kind == "conv":      # layer y = conv(x, w): dgrad gathers with the transposed view, wgrad with the forward geometry
            dg = lambda out, out2: ((p_, w, gt, 32, 32, 9, 32 * 9, out), dict(add1=add1, mask=mask, mask_slope=0.0, out2=out2 if second else None))
            wgeo, wsn, wsc = gf, 32 * 9, 9
        else:                   # layer y = conv_transpose(x, w): dgrad is a plain conv, wgrad has mirrored taps
            dg = lambda out, out2: ((p_, w, gf, 32, 32, 32 * 9, 9, out), dict(add1=add1, mask=mask, mask_slope=0.0, out2=out2 if second else None))
            wgeo, wsn, wsc = gt, 9, 32 * 9
        res = []
        with plain_layer_fused(hip_lib, 7 if kind == "conv" else 8):
            for fused in (False, True):
                out, out2 = torch.zeros(B, H, W, 32, device="cuda"), torch.zeros(B, H, W, 32, device="cuda")
                dw, db = torch.zeros(32, 32, 3, 3, device="cuda"), torch.zeros(32, device="cuda")
                defer = K.DeferredWgrads()
                wg = ((p_, q_, wgeo, 32, 32, dw, wsn, wsc), dict(db=db))
                call = dg(out, out2)
                if fused:
                    assert K.conv_wgrad_fused(call, wg, defer)
                else:
                    K.wgrad(*wg[0], db=db, defer=defer)
                    K.conv(*call[0], **call[1])
                K.flush_wgrads(defer)
                torch.cuda.synchronize()
                res.append((out, out2, dw, db))
        (o_a, o2_a, dw_a, db_a), (o_b, o2_b, dw_b, db_b) = res
        assert torch.equal(o_a, o_b) and torch.equal(o2_a, o2_b), kind            # the halo-tile kernel's arithmetic
        if B == 32:
            assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b), kind
        else:
            assert relerr(dw_b.cpu(), dw_a.cpu()) < 1e-5 and relerr(db_b.cpu(), db_a.cpu()) < 1e-5, kind
        # and against torch on the CPU (the layer's own autograd)
        pc, qc = nchw(p_).double(), nchw(q_).double().requires_grad_(True)
        wc = w.cpu().double().requires_grad_(True)
        y = F.conv2d(qc, wc, None, padding=1) if kind == "conv" else F.conv_transpose2d(qc, wc, None, padding=1)
        (y * pc).sum().backward()
        assert relerr(dw_b.cpu(), wc.grad) < TOL, kind
        assert relerr(db_b.cpu(), pc.sum((0, 2, 3))) < TOL, kind
        want = (qc.grad + nchw(add1).double())
        if second:
            assert relerr(nchw(o2_b), want) < TOL, kind
        assert relerr(nchw(o_b), want * (nchw(mask) > 0)) < TOL, kind


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 32, 44])
def test_resfft_block_tail_one_launch(hip_lib, B):
    """mtd_resfft_block_tail (arch/Ours/networks.py:32-36): img = relu(conv3x3(x) + b) and out = x + img + irfft_rows(T) in
    ONE launch of the halo-tile kernel against (a) the two launches it replaces and (b) torch on the CPU: conv2d + the c2r
    row transform of torch.fft (ortho scale 1/8 per direction, imaginary parts of columns 0 and 32 ignored -- T is given
    NON-Hermitian values there, as the post-ReLU spectrum of the block is).  B = 8: the smallest launch the halo-tile
    kernel takes (128 tiles); 32: two tiles per workgroup; 44: 704 tiles on 256 workgroups (two and three tiles, so the
    third tile's halo lands in a buffer the epilogue of the first has read x + irfft from)."""
    from mtd_gan_amd import kernels as K
    H = W = 64
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, 32, generator=gen).cuda()
    T = torch.randn(B, 33, 64, 64, generator=gen).cuda()          # [b][kw][h][re 32 | im 32]
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.1).cuda()
    bias = (torch.randn(32, generator=gen) * 0.1).cuda()
    g = K.geom_fwd(B, H, W, 3, 1, 1)
    img_a, out_a = torch.zeros(B, H, W, 32, device="cuda"), torch.zeros(B, H, W, 32, device="cuda")
    K.conv(x, w, g, 32, 32, 32 * 9, 9, img_a, bias=bias, act=K.ACT_RELU)
    K.irfft_rows(T, out_a, add1=x, add2=img_a)
    img_b, out_b = torch.zeros(B, H, W, 32, device="cuda"), torch.zeros(B, H, W, 32, device="cuda")
    assert K.block_tail_ok(x, w, g, img_b, bias)
    K.block_tail(x, w, g, T, img_b, out_b, bias=bias, act=K.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(img_a, img_b)                               # the same MFMA sums and epilogue arithmetic
    assert relerr(out_b.cpu(), out_a.cpu()) < 1e-5                 # (the two row transforms associate differently)
    # torch on the CPU
    xc, Tc = nchw(x).double(), T.cpu().double()
    spec = torch.complex(Tc[..., :32], Tc[..., 32:]).permute(0, 3, 2, 1)          # (B, c, h, kw)
    rows = torch.fft.irfft(spec, n=64, dim=-1, norm="ortho")                        # (B, c, h, w)
    img = F.relu(F.conv2d(xc, w.cpu().double(), bias.cpu().double(), padding=1))
    assert relerr(nchw(img_b), img) < TOL
    assert relerr(nchw(out_b), xc + img + rows) < TOL
    # shapes outside the halo-tile kernel's domain are refused, nothing is launched
    small = torch.zeros(2, H, W, 32, device="cuda")
    assert not K.block_tail_ok(small, w, K.geom_fwd(2, H, W, 3, 1, 1), small, bias)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 20, 32])
def test_fused_c32_backward_launch_with_row_transform(hip_lib, B):
    """mtd_conv_c32_bwd_irfft: the fused data + weight gradient launch of a block's 3x3 conv also takes the closing row
    transform of the block's backward pass, gx = (dgrad(gm) + g + irfft_rows(gT)) * (x > 0), against the two launches it
    replaces (mtd_conv_c32_bwd, then mtd_irfft_rows with the data gradient as its add) and against torch on the CPU."""
    from mtd_gan_amd import kernels as K
    H = W = 64
    gen = torch.Generator().manual_seed(13)
    r = lambda *s: torch.randn(*s, generator=gen).cuda()
    gm, x, g = r(B, H, W, 32), r(B, H, W, 32), r(B, H, W, 32)
    gT = r(B, 33, 64, 64)
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.1).cuda()
    gd, gf = K.geom_dgrad_s1(B, H, W, 3, 1), K.geom_fwd(B, H, W, 3, 1, 1)
    res = []
    for fused in (False, True):
        dw, db = torch.zeros(32, 32, 3, 3, device="cuda"), torch.zeros(32, device="cuda")
        gx = torch.zeros(B, H, W, 32, device="cuda")
        defer = K.DeferredWgrads()
        wg = ((gm, x, gf, 32, 32, dw, 32 * 9, 9), dict(db=db))
        if fused:
            dg = ((gm, w, gd, 32, 32, 9, 32 * 9, gx), dict(add1=g, mask=x, mask_slope=0.0))
            assert K.conv_wgrad_fusable(dg, wg)
            assert K.conv_wgrad_fused(dg, wg, defer, spec=gT)
        else:
            d1 = torch.zeros(B, H, W, 32, device="cuda")
            with plain_layer_fused(hip_lib, 7):
                assert K.conv_wgrad_fused(((gm, w, gd, 32, 32, 9, 32 * 9, d1), dict(add1=g)), wg, defer)
            K.irfft_rows(gT, gx, add1=d1, mask=x)
        K.flush_wgrads(defer)
        torch.cuda.synchronize()
        res.append((gx, dw, db))
    (gx_a, dw_a, db_a), (gx_b, dw_b, db_b) = res
    assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)       # the weight-gradient role is untouched
    assert relerr(gx_b.cpu(), gx_a.cpu()) < 1e-5
    # torch on the CPU: conv's input gradient + g + c2r row transform, masked
    spec = torch.complex(gT.cpu().double()[..., :32], gT.cpu().double()[..., 32:]).permute(0, 3, 2, 1)
    rows = torch.fft.irfft(spec, n=64, dim=-1, norm="ortho")
    xin = torch.zeros(B, 32, H, W, dtype=torch.double, requires_grad=True)
    (F.conv2d(xin, w.cpu().double(), None, padding=1) * nchw(gm).double()).sum().backward()
    want = (xin.grad + nchw(g).double() + rows) * (nchw(x) > 0)
    assert relerr(nchw(gx_b), want) < TOL


WINO_CASES = [
    # B, Cin, Cout, H, W, what
    (2, 64, 64, 64, 64, "fwd"),        # one tile row of 32 tiles per workgroup, BN = 64
    (2, 64, 128, 16, 16, "fwd"),       # four tile rows per workgroup, BN = 128 (or 64 + more workgroups)
    (3, 256, 256, 8, 8, "fwd"),        # 48 tiles: ragged last workgroup (two images per workgroup), split-K
    (2, 128, 64, 32, 32, "dgrad"),     # data gradient: mirrored taps, transposed weight view
    (5, 512, 512, 8, 8, "dgrad"),      # 80 tiles, deep K with split-K slabs + the shared split-K epilogue
    (1, 80, 192, 10, 12, "fwd"),       # C a multiple of 16 only, odd tile counts
    (6, 512, 512, 4, 4, "fwd"),        # the deep levels: four tiles per image, mostly padding; split-K
    (7, 512, 1024, 2, 2, "dgrad"),     # one tile per image
    (16, 64, 64, 64, 64, "dgrad"),     # 512 workgroups with four K steps: the lean form, two workgroups per CU
    (16, 64, 256, 32, 32, "fwd"),      # C = 64 with wide N: four K steps under BN = 128 (MTD_WINO_NB2_MAXC=64: the narrow lean form)
    (3, 128, 128, 8, 8, "dgrad"),      # F(2x4): two tiles per tile row, 24 tiles -- one ragged workgroup
    (1, 64, 64, 6, 20, "fwd"),         # F(2x4): width a multiple of 4 but not of 8, height not a multiple of 4
]


def _set_wino_split(K, on):
    """mtd_set_option("wino_split", on): the split-bf16 Winograd kernel (csrc/conv_winograd_split.h) or the fp32 MFMA one."""
    from mtd_gan_amd import _lib
    assert _lib.lib().mtd_set_option(b"wino_split", int(on)) == 0
    K._wino_px_cache.clear()
    K._igemm_ws_cache.clear()
    K.weights_changed(None)


@pytest.mark.gpu
@pytest.mark.parametrize("pipe", ["bf16x3", "fp32"])
@pytest.mark.parametrize("form", ["plan", "f2x2"])
@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_vs_torch(hip_lib, case, form, pipe):
    """csrc/conv_winograd.hip: forward conv and data gradient with the full epilogue (two scales by batch half, bias, two adds,
    LeakyReLU, mask) against torch on the CPU in float64 and against the implicit GEMM on the same inputs (the two differ by
    fp32 rounding only: 2e-5).  form "plan": what the library picks per layer -- F(2x4, 3x3) on maps at least 8 wide whose width
    is a multiple of 4, F(2x2, 3x3) elsewhere; "f2x2": F(2x2, 3x3) everywhere (mtd_conv_winograd_f4_min_w(0)).
    pipe "bf16x3" (round 5; mtd_set_option("wino_split", 1)): the products on the bf16 matrix pipe from exact three-way splits of both operands, six
    per fp32 product (csrc/conv_winograd_split.h) -- held to the SAME bounds as the fp32 MFMA kernel ("fp32", the default),
    and its error against float64 may not exceed twice the fp32 kernel's."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, W, what = case
    if form == "f2x2" and not (W % 4 == 0 and W >= 8):
        pytest.skip("the plan takes F(2x2, 3x3) for this map anyway")
    old_min_w = K.winograd_f4_min_w(0 if form == "f2x2" else 8)
    _set_wino_split(K, pipe == "bf16x3")
    try:
        _winograd_conv_case(K, B, Ci, Co, H, W, what, form, pipe)
    finally:
        K.winograd_f4_min_w(old_min_w)
        _set_wino_split(K, False)


def _winograd_conv_case(K, B, Ci, Co, H, W, what, form, pipe="bf16x3"):
    gen = torch.Generator().manual_seed(17)
    r = lambda *s: torch.randn(*s, generator=gen)
    x = r(B, H, W, Ci if what == "fwd" else Co).cuda()
    w = (r(Co, Ci, 3, 3) * (9 * Ci) ** -0.5).cuda()
    N, Cc = (Co, Ci) if what == "fwd" else (Ci, Co)
    bias, add1, add2, mask = (r(N) * 0.1).cuda(), r(B, H, W, N).cuda(), r(B, H, W, N).cuda(), r(B, H, W, N).cuda()
    s1, s2 = torch.tensor([0.7], device="cuda"), torch.tensor([1.3], device="cuda")
    split = (B // 2) * H * W if B > 1 else 0
    if what == "fwd":
        geom, wsn, wsc = K.geom_fwd(B, H, W, 3, 1, 1), Ci * 9, 9
    else:
        geom, wsn, wsc = K.geom_dgrad_s1(B, H, W, 3, 1), 9, Ci * 9
    kw = dict(scale=s1, bias=bias, add1=add1, add2=add2, act=K.ACT_LRELU, mask=mask, mask_slope=0.2)
    if split:
        kw.update(scale2=s2, scale_split=split)
    assert K.winograd_takes(geom, N, Cc, kw)
    assert K.winograd_patch_w(geom, N, Cc) == (6 if (form == "plan" and W % 4 == 0 and W >= 8) else 4) + (16 if pipe == "bf16x3" else 0)
    outs = []
    for wino in (True, False):
        K.WINOGRAD = wino
        try:
            out = torch.zeros(B, H, W, N, device="cuda")
            K.conv(x, w, geom, N, Cc, wsn, wsc, out, **kw)
            torch.cuda.synchronize()
            outs.append(out)
        finally:
            K.WINOGRAD = True
    assert relerr(outs[0].cpu(), outs[1].cpu()) < 2e-5
    xc, wc = nchw(x).double(), w.cpu().double()
    y = F.conv2d(xc, wc, None, padding=1) if what == "fwd" else F.conv_transpose2d(xc, wc, None, padding=1)
    sc = torch.full((B, 1, 1, 1), 0.7, dtype=torch.double)
    if split:
        sc[B // 2:] = 1.3
    v = y * sc + bias.cpu().double().view(1, -1, 1, 1) + nchw(add1).double() + nchw(add2).double()
    v = F.leaky_relu(v, 0.2) * torch.where(nchw(mask) > 0, 1.0, 0.2)
    assert relerr(nchw(outs[0]), v) < TOL
    if pipe == "bf16x3":
        # fp32 accuracy, not merely parity: against float64 no worse than twice the fp32 MFMA kernel on the same inputs
        _set_wino_split(K, False)
        out32 = torch.zeros(B, H, W, N, device="cuda")
        K.conv(x, w, geom, N, Cc, wsn, wsc, out32, **kw)
        torch.cuda.synchronize()
        _set_wino_split(K, True)
        e3 = (nchw(outs[0]).double() - v).abs().max().item() / v.abs().max().item()
        e32 = (nchw(out32).double() - v).abs().max().item() / v.abs().max().item()
        assert e3 <= 2 * e32 + 2e-7, (e3, e32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 128, 128, "fwd", "relu_add"), (2, 128, 132, "dgrad", "relu"), (1, 256, 256, "fwd", "relu"),
                                  (3, 128, 128, "dgrad", "lrelu_mask"), (1, 256, 256, "dgrad", "relu_noadd"), (3, 136, 132, "fwd", "none"),
                                  (5, 130, 140, "fwd", "lrelu"), (8, 512, 512, "fwd", "relu_add")])
def test_winograd_32_channel_form(hip_lib, case):
    """The generator's 32 -> 32 layers on whole-slice maps (side >= kernels.WINO_C32_MIN_HW = 128) run on the F(2x4, 3x3)
    form: the persistent kernel of csrc/conv_wino_c32.h (weights in registers, two half-steps per 256-pixel block; one residual
    operand at most) or, with a mask, the general kernel's 32-channel workgroups (wino_conv_kernel<1, false, 6>).  Forward conv /
    ConvTranspose gather with the epilogues the inference schedule uses -- relu(conv + b) + x (MTD_ACT_RELU_ADD),
    relu(conv + b + skip), relu(conv + b) -- and the others, on whole blocks, ragged last blocks (136 x 132 x 3, 130 x 140 x 5) and
    the bench's own size, against torch on the CPU in float64 and against the halo-tile implicit GEMM on the same inputs.  On a 64 x 64 map the
    same layer stays on the implicit GEMM (the training path is unchanged)."""
    from mtd_gan_amd import kernels as K
    B, H, W, what, epi = case
    gen = torch.Generator().manual_seed(29)
    r = lambda *s: torch.randn(*s, generator=gen)
    x = r(B, H, W, 32).cuda()
    w = (r(32, 32, 3, 3) * (9 * 32) ** -0.5).cuda()
    bias, add1, mask = (r(32) * 0.1).cuda(), r(B, H, W, 32).cuda(), r(B, H, W, 32).cuda()
    if what == "fwd":
        geom, wsn, wsc = K.geom_fwd(B, H, W, 3, 1, 1), 32 * 9, 9
    else:
        geom, wsn, wsc = K.geom_dgrad_s1(B, H, W, 3, 1), 9, 32 * 9
    kw = {"relu_add": dict(bias=bias, add1=add1, act=K.ACT_RELU_ADD), "relu": dict(bias=bias, add1=add1, act=K.ACT_RELU),
          "lrelu_mask": dict(bias=bias, add1=add1, act=K.ACT_LRELU, mask=mask, mask_slope=0.2),
          "relu_noadd": dict(bias=bias, act=K.ACT_RELU), "none": dict(add1=add1), "lrelu": dict(bias=bias, act=K.ACT_LRELU)}[epi]
    assert K.winograd_takes(geom, 32, 32, kw) and K.winograd_patch_w(geom, 32, 32) == 6
    assert not K.winograd_takes(K.geom_fwd(B, 64, 64, 3, 1, 1), 32, 32, kw)
    if epi == "relu_add":
        assert K.conv_relu_add_ok(x, w, geom, 32, 32, wsn, wsc, torch.empty_like(x), bias=bias, add1=add1)
    outs = []
    for wino in (True, False):
        K.WINOGRAD = wino
        try:
            out = torch.zeros(B, H, W, 32, device="cuda")
            K.FLOP_COUNT = {}
            try:
                K.conv(x, w, geom, 32, 32, wsn, wsc, out, **kw)
            finally:
                fc, K.FLOP_COUNT = K.FLOP_COUNT, None
            torch.cuda.synchronize()
            assert (fc.get("conv_winograd_saved", 0.0) > 0) == wino      # (which kernel took the launch)
            outs.append(out)
        finally:
            K.WINOGRAD = True
    assert relerr(outs[0].cpu(), outs[1].cpu()) < 2e-5
    xc, wc = nchw(x).double(), w.cpu().double()
    y = F.conv2d(xc, wc, None, padding=1) if what == "fwd" else F.conv_transpose2d(xc, wc, None, padding=1)
    if "bias" in kw:
        y = y + bias.cpu().double().view(1, -1, 1, 1)
    a1 = nchw(add1).double()
    if epi == "relu_add":
        v = F.relu(y) + a1
    elif epi == "relu":
        v = F.relu(y + a1)
    elif epi == "relu_noadd":
        v = F.relu(y)
    elif epi == "none":
        v = y + a1
    elif epi == "lrelu":
        v = F.leaky_relu(y, 0.2)
    else:
        v = F.leaky_relu(y + a1, 0.2) * torch.where(nchw(mask) > 0, 1.0, 0.2)
    assert relerr(nchw(outs[0]), v) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 64, 64, "fwd", True, True), (32, 64, 64, "dgrad", False, True), (3, 64, 64, "dgrad", True, False),
                                  (2, 32, 48, "fwd", False, True), (5, 40, 36, "dgrad", True, True)])
def test_winograd_32_channel_masked_two_output_form(hip_lib, monkeypatch, case):
    """csrc/conv_wino_c32.h, MASKED2 (round 6): the persistent F(2x4, 3x3) kernel in the data-gradient role of the generator's plain
    32 -> 32 layers -- out2 = conv (+ add1), out = out2 * (mask > 0 ? 1 : slope) -- against the halo-tile / implicit-GEMM launch of
    the same arguments and against float64; both tap orders (a decoder's data gradient gathers like a forward conv, an encoder's
    like a transposed one), with and without the residual operand and the second output; ragged block counts; repeatable."""
    from mtd_gan_amd import kernels as K
    B, H, W, what, has_add, has_out2 = case
    monkeypatch.setattr(K, "WINO_C32_BWD", True)
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, H, W, 32, generator=gen).cuda()
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).cuda()
    add1 = torch.randn(B, H, W, 32, generator=gen).cuda() if has_add else None
    mask = torch.randn(B, H, W, 32, generator=gen).cuda()
    mask[0, 0, :4] = 0.0                                          # exact zeros take the slope side
    if what == "fwd":
        geom, wsn, wsc = K.geom_fwd(B, H, W, 3, 1, 1), 288, 9
    else:
        geom, wsn, wsc = K.geom_dgrad_s1(B, H, W, 3, 1), 9, 288
    res = {}
    for form in ("wino", "wino_again", "igemm"):
        out = torch.full_like(x, float("nan"))
        # (the halo-tile kernel writes a second output on the generator's own shape only: kernels.fuses_masked_cotangent)
        out2 = torch.full_like(x, float("nan")) if has_out2 and (form != "igemm" or K.fuses_masked_cotangent(B, H, W, 32, 32)) else None
        kw = dict(add1=add1, mask=mask, mask_slope=0.0, out2=out2)
        if form != "igemm":
            kw["wino32"] = True
        K.FLOP_COUNT = {}
        try:
            K.conv(x, w, geom, 32, 32, wsn, wsc, out, **kw)
        finally:
            fc, K.FLOP_COUNT = K.FLOP_COUNT, None
        torch.cuda.synchronize()
        assert (fc.get("conv_winograd_saved", 0.0) > 0) == (form != "igemm")
        assert not torch.isnan(out).any() and (out2 is None or not torch.isnan(out2).any())
        res[form] = (out, out2)
    assert torch.equal(res["wino"][0], res["wino_again"][0])
    xc, wc = nchw(x).double(), w.cpu().double()
    y = F.conv2d(xc, wc, None, padding=1) if what == "fwd" else F.conv_transpose2d(xc, wc, None, padding=1)
    if has_add:
        y = y + nchw(add1).double()
    ym = y * (nchw(mask) > 0).double()
    assert relerr(nchw(res["wino"][0]), ym) < TOL and relerr(res["wino"][0].cpu(), res["igemm"][0].cpu()) < 2e-5
    if has_out2:
        assert torch.equal(res["wino"][1], res["wino_again"][1])
        assert relerr(nchw(res["wino"][1]), y) < TOL
        if res["igemm"][1] is not None:
            assert relerr(res["wino"][1].cpu(), res["igemm"][1].cpu()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("nprob", [2, 3])
@pytest.mark.parametrize("case", [(64, 1024, 512, 2, "fwd"), (64, 512, 512, 4, "fwd"), (64, 1024, 256, 8, "fwd"), (64, 256, 256, 16, "dgrad"),
                                  (64, 512, 1024, 4, "dgrad"), (64, 128, 128, 32, "dgrad"), (32, 64, 64, 64, "fwd"), (64, 128, 1, 64, "fwd")])
def test_conv_pair_equals_two_launches(hip_lib, monkeypatch, case, nprob):
    """kernels.conv_pair / mtd_conv_winograd_pair (round 6): two convs of one shape -- the mirror layers of the discriminator's two
    decoders -- in ONE launch of wino_conv_multi_kernel (+ one slab-sum launch for both) against two conv() calls, for the forward form
    (bias, LeakyReLU, per-half 1/sigma) and the data-gradient form (mask, two scales), from the 2x2 to the 64x64 maps.  The pair's
    split of K is planned for its whole grid (half the slices per problem): the same products in another grouping of the K sum -- equal
    to the single launches within 1e-5 of the tensor's largest element (1e-4 under the element-wise measure of tests/_metrics.py; K up to 9 216 products), bit for bit where neither splits K; the pair itself is
    repeatable.  The last case (one output channel) is not a Winograd layer: conv_pair falls back to two launches."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, what = case
    gen = torch.Generator().manual_seed(53)
    sc = [torch.tensor([0.8], device="cuda"), torch.tensor([1.3], device="cuda")]
    calls, outs = [], {}
    for form in ("pair", "pair again", "single"):
        gen.manual_seed(53)
        lst = []
        for i in range(nprob):
            x = torch.randn(B, H, H, Ci, generator=gen).cuda()
            w = (torch.randn(Co, Ci, 3, 3, generator=gen) * (9 * Ci) ** -0.5).cuda()
            out = torch.full((B, H, H, Co), float("nan"), device="cuda")
            if what == "fwd":
                bias = torch.randn(Co, generator=gen).cuda()
                lst.append(((x, w, K.geom_fwd(B, H, H, 3, 1, 1), Co, Ci, Ci * 9, 9, out),
                            dict(bias=bias, act=K.ACT_LRELU, scale=sc[i % 2], scale2=sc[1 - i % 2], scale_split=(B // 2) * H * H)))
            else:
                mask = torch.randn(B, H, H, Co, generator=gen).cuda()
                wt = (torch.randn(Ci, Co, 3, 3, generator=gen) * (9 * Ci) ** -0.5).cuda()       # the layer's OIHW weight: Ci outputs, Co inputs
                lst.append(((x, wt, K.geom_dgrad_s1(B, H, H, 3, 1), Co, Ci, 9, Co * 9, out),
                            dict(add1=None, mask=mask, mask_slope=0.2, scale=sc[i % 2], scale2=sc[1 - i % 2], scale_split=(B // 2) * H * H)))
        K.FLOP_COUNT = {}
        try:
            if form != "single":
                K.conv_group(lst)                  # (two or three problems: kernels.conv_pair is the group of two)
            else:
                for args, kw in lst:
                    K.conv(*args, **kw)
        finally:
            fc, K.FLOP_COUNT = K.FLOP_COUNT, None
        torch.cuda.synchronize()
        outs[form] = ([c[0][7] for c in lst], fc)
    for a, a2, b in zip(outs["pair"][0], outs["pair again"][0], outs["single"][0]):
        assert not torch.isnan(a).any() and torch.equal(a, a2)
        assert relerr(a.cpu(), b.cpu()) < 1e-4 and (a - b).abs().max().item() < 1e-5 * b.abs().max().item()
        if B * H * H >= 131072:          # (a grid this large is not split either way: the very same sums)
            assert torch.equal(a, b)
    # the pair is ONE launch where the layer is a Winograd layer (the accounting says so), two otherwise
    assert outs["single"][1]["launches"] == nprob
    assert outs["pair"][1]["launches"] == (nprob if Co == 1 else 1)
    assert outs["pair"][1].get("conv_mfma", 0.0) == pytest.approx(outs["single"][1].get("conv_mfma", 0.0))


@pytest.mark.gpu
def test_winograd_32_channel_form_is_repeatable(hip_lib):
    """The persistent kernel's pipeline hands three blocks' work between its waves through LDS (two input images, one exchange
    image, two barriers per block): the same launch twice gives the same bits, on a whole-block shape, a ragged one and the
    training patches' shape (where the generator's forward pass asks for it: conv(..., wino32=True)), every element written."""
    from mtd_gan_amd import kernels as K
    for (B, H, W) in ((32, 64, 64), (2, 128, 128), (3, 136, 132)):
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(B, H, W, 32, generator=gen).cuda()
        w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.06).cuda()
        bias, add1 = torch.randn(32, generator=gen).cuda(), torch.randn(B, H, W, 32, generator=gen).cuda()
        geom = K.geom_fwd(B, H, W, 3, 1, 1)
        ref = None
        for _ in range(5):
            out = torch.full_like(x, float("nan"))
            K.FLOP_COUNT = {}
            try:
                K.conv(x, w, geom, 32, 32, 288, 9, out, bias=bias, add1=add1, act=K.ACT_RELU, wino32=True)
            finally:
                fc, K.FLOP_COUNT = K.FLOP_COUNT, None
            torch.cuda.synchronize()
            assert fc.get("conv_winograd_saved", 0.0) > 0
            assert not torch.isnan(out).any()
            if ref is None:
                ref = out
            else:
                assert torch.equal(ref, out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 64, 64, 64), (3, 128, 64, 16, 16), (5, 256, 512, 8, 8), (2, 64, 128, 10, 12), (32, 64, 64, 32, 32),
                                  (3, 64, 64, 6, 20), (7, 128, 192, 8, 8)])
def test_winograd_weight_gradient_vs_torch(hip_lib, case):
    """csrc/conv_wgrad_wino.h: Winograd F(2x2, 3x3) weight + bias gradient (both operands transformed) against torch's autograd on
    the CPU in float64 and against the row-window / block-window kernels on the same inputs; with and without accumulation into an
    existing gradient.  (5, 256, 512, 8, 8): ragged slices; (2, 64, 128, 10, 12), (3, 64, 64, 6, 20): odd tile counts;
    (7, 128, 192, 8, 8): 14 tiles, C = 192.  (The F(2x4, 3x3) form this test also covered in rounds 4-5 was removed in round 6.)"""
    from mtd_gan_amd import kernels as K
    _winograd_wgrad_case(K, case, "f2x2")


def _winograd_wgrad_case(K, case, form):
    from mtd_gan_amd import _lib
    B, Ci, Co, H, W = case
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(B, H, W, Ci, generator=gen).cuda()
    gy = torch.randn(B, H, W, Co, generator=gen).cuda()
    geom = K.geom_fwd(B, H, W, 3, 1, 1)
    L = _lib.lib()
    L.mtd_conv_wgrad_override.argtypes = [_lib.C.c_int, _lib.C.c_int]
    res = {}
    for name, cfg in (("wino", 16), ("direct", -2)):
        L.mtd_conv_wgrad_override(cfg, -1)
        try:
            dw = torch.full((Co, Ci, 3, 3), 0.5, device="cuda")
            db = torch.full((Co,), -1.0, device="cuda")
            K.wgrad(gy, x, geom, Co, Ci, dw, Ci * 9, 9, db=db, accumulate=True)
            dw2, db2 = torch.empty(Co, Ci, 3, 3, device="cuda"), torch.empty(Co, device="cuda")
            K.wgrad(gy, x, geom, Co, Ci, dw2, Ci * 9, 9, db=db2)
            torch.cuda.synchronize()
            res[name] = (dw, db, dw2, db2)
        finally:
            L.mtd_conv_wgrad_override(-1, -1)
    xc = nchw(x).double()
    wc = torch.zeros(Co, Ci, 3, 3, dtype=torch.double, requires_grad=True)
    bc = torch.zeros(Co, dtype=torch.double, requires_grad=True)
    (F.conv2d(xc, wc, bc, padding=1) * nchw(gy).double()).sum().backward()
    dw, db, dw2, db2 = res["wino"]
    assert relerr(dw2.cpu(), wc.grad) < TOL and relerr(db2.cpu(), bc.grad) < TOL
    assert relerr((dw - 0.5).cpu(), wc.grad) < TOL and relerr((db + 1.0).cpu(), bc.grad) < TOL
    assert relerr(dw2.cpu(), res["direct"][2].cpu()) < 1e-4 and relerr(db2.cpu(), res["direct"][3].cpu()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(4, 64, 64, 64, 64, 3, 1), (6, 128, 64, 16, 16, 3, 1), (10, 256, 512, 8, 8, 3, 1), (2, 64, 128, 10, 12, 3, 1),
                                  (64, 128, 128, 32, 32, 3, 1), (4, 32, 64, 16, 16, 3, 0), (6, 512, 512, 4, 4, 3, 1), (10, 256, 256, 2, 2, 3, 1),
                                  (6, 64, 64, 32, 32, 4, 1), (4, 128, 128, 8, 8, 4, 1), (10, 512, 512, 2, 2, 4, 1), (6, 512, 512, 1, 1, 1, 1),
                                  (4, 96, 32, 6, 6, 3, 1)])
def test_weight_gradient_pair_equals_two_launches(hip_lib, case):
    """mtd_conv_wgrad_pair: the weight gradients of the two halves of one batch from ONE launch of the slab-producing kernel
    (its slices aligned to the halves; discriminator_path.wgrad_sn) against mtd_conv_wgrad on each half and torch's autograd in
    float64; the bias gradient is the accumulated sum of both halves.  (10, 256, 512, 8, 8): 80 tiles per half = ragged
    slices; (2, 64, 128, 10, 12): one image per half, odd tile rows.  case = (B, Cin, Cout, H, W, k, pairs): k = 3 the 3x3
    stride-1 layers (Winograd, block-window and LDS-staged kernels by map size), k = 4 the stride-2 4x4 `down` layers
    (H, W = output size: halo-window and all-taps kernels), k = 1 the bottleneck's 1x1 layers; pairs = 0: the plan is the
    row-window kernel, which has no pair form -- two launches, same results."""
    from mtd_gan_amd import _lib
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, W, k, pairs = case
    Bh = B // 2
    L = _lib.lib()
    default_rule = pairs and k != 1 and H >= 8 and Ci % 64 == 0 and Co % 64 == 0      # Winograd and stride-2 halo-window plans
    for mode, expect in ((1, int(default_rule)), (3, pairs)):
        prev = L.mtd_conv_wgrad_pair_mode(mode)
        try:
            _pair_case(K, _lib, B, Bh, Ci, Co, H, W, k, expect)
        finally:
            L.mtd_conv_wgrad_pair_mode(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(64, 512, 512, 4, 3), (64, 512, 512, 2, 3), (64, 1024, 512, 2, 3), (64, 512, 512, 2, 4), (64, 512, 512, 1, 4),
                                  (64, 512, 512, 1, 1), (64, 256, 512, 4, 4), (8, 128, 96, 4, 3), (64, 64, 64, 16, 3)])
def test_weight_gradient_half_scale_equals_two_scaled_launches(hip_lib, case):
    """mtd_wgrad_args.half_scale (round 6): both halves of a paired discriminator pass in ONE launch of the small-map kernels, the
    cotangent of each half multiplied by that half's own factor as it is used -- against s1 * dw(first half) + s2 * dw(second half) from
    two launches, and the bias gradient against the unscaled sum.  case = (B, Cin, Cout, OUTPUT map side, k): k = 3 the block-window
    kernels (4x4, 2x2 maps), k = 4 the stride-2 layers (all-taps kernel, wgrad_kernel<1, 1, 3>), k = 1 the bottleneck; the last case is
    a large map, whose plan (Winograd) does not take the form: refused, nothing launched."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, k = case
    Bh = B // 2
    stride, pad = (2, 1) if k == 4 else (1, (k - 1) // 2)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, H * stride, H * stride, Ci, generator=gen).cuda()
    gy = torch.randn(B, H, H, Co, generator=gen).cuda()
    geom = K.geom_fwd(B, H * stride, H * stride, k, stride, pad)
    m_first = Bh * H * H
    ok = K.wgrad_half_ok(geom, Co, Ci, m_first)
    if H == 16:
        assert not ok
        return
    assert ok == (m_first % 32 == 0)
    if not ok:
        return
    s1, s2 = torch.tensor([1.7], device="cuda"), torch.tensor([0.45], device="cuda")
    dw = torch.full((Co, Ci, k, k), float("nan"), device="cuda")
    db = torch.zeros(Co, device="cuda")
    K.wgrad(gy, x, geom, Co, Ci, dw, Ci * k * k, k * k, db=db, accumulate=False, accumulate_bias=True, half=(s1, s2, m_first))
    parts, dbs = [], []
    for lo, hi in ((0, Bh), (Bh, B)):
        d = torch.empty(Co, Ci, k, k, device="cuda")
        b = torch.empty(Co, device="cuda")
        K.wgrad(gy[lo:hi], x[lo:hi], K.mtd_geom_with_batch(geom, hi - lo), Co, Ci, d, Ci * k * k, k * k, db=b)
        parts.append(d)
        dbs.append(b)
    torch.cuda.synchronize()
    ref = 1.7 * parts[0].double() + 0.45 * parts[1].double()
    assert relerr(dw.cpu(), ref.cpu()) < 2e-5
    assert relerr(db.cpu(), (dbs[0].double() + dbs[1].double()).cpu()) < 2e-5
    # accumulation into an existing gradient, and repeatability
    dw2 = torch.full_like(dw, 0.25)
    K.wgrad(gy, x, geom, Co, Ci, dw2, Ci * k * k, k * k, db=None, accumulate=True, half=(s1, s2, m_first))
    dw3 = torch.full_like(dw, float("nan"))
    K.wgrad(gy, x, geom, Co, Ci, dw3, Ci * k * k, k * k, db=None, accumulate=False, half=(s1, s2, m_first))
    torch.cuda.synchronize()
    assert torch.equal(dw3, dw) and relerr((dw2 - 0.25).cpu(), ref.cpu()) < 2e-5


def _pair_case(K, _lib, B, Bh, Ci, Co, H, W, k, pairs):
    gen = torch.Generator().manual_seed(29)
    stride, pad = (2, 1) if k == 4 else (1, (k - 1) // 2)
    x = torch.randn(B, H * stride, W * stride, Ci, generator=gen).cuda()
    gy = torch.randn(B, H, W, Co, generator=gen).cuda()
    geom = K.geom_fwd(B, H * stride, W * stride, k, stride, pad)
    kk = k * k
    d1, d2 = torch.full((Co, Ci, k, k), 7.0, device="cuda"), torch.full((Co, Ci, k, k), 7.0, device="cuda")
    a = K.WgradArgs()
    a.g = geom
    a.p, a.p_ld, a.N, a.q, a.q_ld, a.C = gy.data_ptr(), Co, Co, x.data_ptr(), Ci, Ci
    a.dw, a.w_sn, a.w_sc = d1.data_ptr(), Ci * kk, kk
    ok = _lib.lib().mtd_conv_wgrad_pair_ok(_lib.C.byref(a), Bh)                 # 2: the plan also adds a second cotangent on load
    assert (ok != 0) == bool(pairs)
    db = torch.full((Co,), -1.0, device="cuda")
    K.wgrad_pair(gy, x, geom, Bh, Co, Ci, d1, d2, Ci * kk, kk, db=db, accumulate_bias=True)
    r1, r2 = torch.empty_like(d1), torch.empty_like(d2)
    rb = torch.zeros(Co, device="cuda")
    gh = K.geom_fwd(Bh, H * stride, W * stride, k, stride, pad)
    K.wgrad(gy[:Bh], x[:Bh], gh, Co, Ci, r1, Ci * kk, kk, db=rb, accumulate=False, accumulate_bias=True)
    K.wgrad(gy[Bh:], x[Bh:], gh, Co, Ci, r2, Ci * kk, kk, db=rb, accumulate=False, accumulate_bias=True)
    # a second cotangent (the decoders' sums over two task passes): gradients of gy + gz, added inside the launch where ok == 2,
    # by kernels.add otherwise -- against the two-launch form on the materialised sum
    gz = torch.randn(B, H, W, Co, generator=gen).cuda()
    e1, e2 = torch.empty_like(d1), torch.empty_like(d2)
    eb = torch.zeros(Co, device="cuda")
    K.wgrad_pair(gy, x, geom, Bh, Co, Ci, e1, e2, Ci * kk, kk, db=eb, accumulate_bias=True, p_add=gz)
    gs = gy + gz
    s1, s2 = torch.empty_like(d1), torch.empty_like(d2)
    sb = torch.zeros(Co, device="cuda")
    K.wgrad(gs[:Bh], x[:Bh], gh, Co, Ci, s1, Ci * kk, kk, db=sb, accumulate=False, accumulate_bias=True)
    K.wgrad(gs[Bh:], x[Bh:], gh, Co, Ci, s2, Ci * kk, kk, db=sb, accumulate=False, accumulate_bias=True)
    torch.cuda.synchronize()
    assert relerr(e1.cpu(), s1.cpu()) < 2e-5 and relerr(e2.cpu(), s2.cpu()) < 2e-5 and relerr(eb.cpu(), sb.cpu()) < 2e-5
    assert relerr(d1.cpu(), r1.cpu()) < 2e-5 and relerr(d2.cpu(), r2.cpu()) < 2e-5 and relerr((db + 1.0).cpu(), rb.cpu()) < 2e-5
    for half, got in ((slice(0, Bh), d1), (slice(Bh, B), d2)):
        wc = torch.zeros(Co, Ci, k, k, dtype=torch.double, requires_grad=True)
        (F.conv2d(nchw(x[half]).double(), wc, None, stride=stride, padding=pad) * nchw(gy[half]).double()).sum().backward()
        assert relerr(got.cpu(), wc.grad) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("B,second", [(4, False), (12, True), (20, True), (40, True)])
def test_fused_c32_backward_counted_waits_same_bits(hip_lib, monkeypatch, B, second):
    """conv_c32_bwd.hip hands a 32-pixel block's halo buffer to the next DMA after `s_waitcnt vmcnt(4 | 8)` -- counts that assume
    the epilogue issues exactly that many vector-memory instructions after the DMAs.  mtd_set_option("c32f_safe_wait", 1) waits for everything
    instead; both must give the same bits, with and without the second output, on grids below 256 workgroups (B = 4: 64 tiles),
    with one and two live tiles per workgroup (B = 20), and with several iterations per workgroup (B = 40: 640 tiles)."""
    from mtd_gan_amd import _lib, kernels as K
    H = W = 64
    gen = torch.Generator().manual_seed(29)
    r = lambda *s: torch.randn(*s, generator=gen).cuda()
    p_, q_, add1, mask = r(B, H, W, 32), r(B, H, W, 32), r(B, H, W, 32), r(B, H, W, 32)
    w = (torch.randn(32, 32, 3, 3, generator=gen) * 0.1).cuda()
    gf, gt = K.geom_fwd(B, H, W, 3, 1, 1), K.geom_dgrad_s1(B, H, W, 3, 1)
    res = []
    for safe in ("0", "1"):
        assert _lib.lib().mtd_set_option(b"c32f_safe_wait", int(safe)) == 0
        out, out2 = torch.zeros(B, H, W, 32, device="cuda"), torch.zeros(B, H, W, 32, device="cuda")
        dw, db = torch.zeros(32, 32, 3, 3, device="cuda"), torch.zeros(32, device="cuda")
        defer = K.DeferredWgrads()
        call = ((p_, w, gt, 32, 32, 9, 32 * 9, out), dict(add1=add1, mask=mask, mask_slope=0.0, out2=out2 if second else None))
        with plain_layer_fused(hip_lib, 7):
            assert K.conv_wgrad_fused(call, ((p_, q_, gf, 32, 32, dw, 32 * 9, 9), dict(db=db)), defer)
        K.flush_wgrads(defer)
        torch.cuda.synchronize()
        res.append((out, out2, dw, db))
    assert _lib.lib().mtd_set_option(b"c32f_safe_wait", 0) == 0
    import ctypes
    v = ctypes.c_int(-1)
    assert _lib.lib().mtd_get_option(b"c32f_safe_wait", ctypes.byref(v)) == 0 and v.value == 0
    assert _lib.lib().mtd_set_option(b"no_such_option", 1) != 0
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(3, 64, 64, 32, 32), (2, 64, 128, 64, 64), (5, 128, 64, 16, 16), (2, 256, 256, 16, 16), (3, 64, 64, 20, 28),
                                  (2, 512, 64, 16, 16), (2, 64, 64, 18, 18), (26, 64, 64, 64, 64), (64, 128, 128, 32, 32)])
def test_stride2_winograd_f33_22_vs_float64(hip_lib, case):
    """F(3x3, 2x2) form of the 4x4 / stride-2 / padding-1 layers (csrc/conv_wino_s2.h; networks.py:185-215 down1..3): the forward
    conv (four phases of the padded input in the K sum, ragged last tiles, full epilogue, split-K on the small grids) and the four-parity data gradient
    as one grid (with a residual operand and the activation mask), against float64; the error may not exceed twice the implicit
    GEMM's on the same operands (and 1e-3 in any case).  (64, 128, 128, 32, 32) = down2 at the D step's 64 images: the plan's
    128-channel workgroups with a three-way split of K; (26, 64, 64, 64, 64): 392 workgroups of four K steps = the two-workgroups-per-CU
    form of the data gradient."""
    from mtd_gan_amd import kernels as K
    B, Ci, Co, H, W = case
    x = rnd(B, Ci, H, W, seed=31)
    w = rnd(Co, Ci, 4, 4, seed=32, scale=0.05)
    b = rnd(Co, seed=33)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=1) * 0.7 + b.double().view(1, -1, 1, 1), 0.2)
    OH, OW = H // 2, W // 2
    xd, wd, bd = nhwc(x), w.cuda(), b.cuda()
    scale = torch.tensor([0.7], device="cuda")
    geom = K.geom_fwd(B, H, W, 4, 2, 1)
    assert K.winograd_s2_takes(geom, Co, Ci, {})
    outs = {}
    for form in ("igemm", "wino"):
        K.WINO_S2 = 2 if form == "wino" else 0
        out = torch.full((B, OH, OW, Co), float("nan"), device="cuda")
        K.conv(xd, wd, geom, Co, Ci, Ci * 16, 16, out, scale=scale, bias=bd, act=K.ACT_LRELU)
        outs[form] = relerr(nchw(out).double(), ref)
    K.WINO_S2 = 1
    assert outs["wino"] < max(TOL * 1e-2, 2 * outs["igemm"]), outs
    # data gradient: dx = add1 + conv_transpose(cot, w), masked by (mask > 0 ? 1 : 0.2), four classes in one grid
    cot = rnd(B, Co, OH, OW, seed=34)
    add = rnd(B, Ci, H, W, seed=35)
    msk = rnd(B, Ci, H, W, seed=36)
    want = (F.conv_transpose2d(cot.double(), w.double(), None, stride=2, padding=1) * 0.7 + add.double()) * torch.where(msk > 0, 1.0, 0.2).double()
    cd, addd, mskd = nhwc(cot), nhwc(add), nhwc(msk)
    errs = {}
    for form in ("igemm", "wino"):
        K.WINO_S2 = 2 if form == "wino" else 0
        dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        calls = []
        for py in range(2):
            for px in range(2):
                gp = K.geom_dgrad_s2(B, H, W, py, px)
                calls.append(((cd, wd, gp, Ci, Co, 16, Ci * 16, dx), dict(scale=scale, add1=addd, mask=mskd, mask_slope=0.2)))
        K.conv_multi(calls)
        errs[form] = relerr(nchw(dx).double(), want)
    K.WINO_S2 = 1
    assert errs["wino"] < max(TOL * 1e-2, 2 * errs["igemm"]), errs


@pytest.mark.parametrize("case", [(3, 64, 64, 32, 32), (2, 64, 128, 16, 16), (5, 128, 64, 8, 8), (3, 64, 64, 10, 14), (2, 128, 128, 17, 9), (16, 64, 64, 32, 32)])
def test_stride2_winograd_weight_gradient_vs_float64(hip_lib, case):
    """F(3x3, 2x2) weight gradient of the 4x4 / stride-2 / padding-1 layers (csrc/conv_wgrad_wino_s2.h; networks.py:185-215): the four
    phases as 4 C channels, ragged last tiles, the bias gradient, accumulation -- against autograd in float64, with the error of
    the halo-window / LDS-staged kernels on the same operands as the yardstick (at most twice theirs, and 1e-3 in any case).
    case = (B, Cin, Cout, OH, OW)."""
    from mtd_gan_amd import _lib
    from mtd_gan_amd import kernels as K
    B, Ci, Co, OH, OW = case
    H, W = 2 * OH, 2 * OW
    x = rnd(B, Ci, H, W, seed=41)
    gy = rnd(B, Co, OH, OW, seed=42)
    wc = torch.zeros(Co, Ci, 4, 4, dtype=torch.double, requires_grad=True)
    bc = torch.zeros(Co, dtype=torch.double, requires_grad=True)
    (F.conv2d(x.double(), wc, bc, stride=2, padding=1) * gy.double()).sum().backward()
    xd, gd = nhwc(x), nhwc(gy)
    geom = K.geom_fwd(B, H, W, 4, 2, 1)
    L = _lib.lib()
    errs = {}
    for name, cfg in (("wino_s2", 18), ("other", -2)):
        L.mtd_conv_wgrad_override(cfg, -1)
        try:
            dw = torch.full((Co, Ci, 4, 4), float("nan"), device="cuda")
            db = torch.full((Co,), float("nan"), device="cuda")
            K.wgrad(gd, xd, geom, Co, Ci, dw, Ci * 16, 16, db=db)
            if name == "wino_s2":
                a = K.WgradArgs()
                a.g = geom
                a.p, a.p_ld, a.N, a.q, a.q_ld, a.C = gd.data_ptr(), Co, Co, xd.data_ptr(), Ci, Ci
                a.dw, a.w_sn, a.w_sc = dw.data_ptr(), Ci * 16, 16
                assert L.mtd_conv_wgrad_plan_cfg(_lib.C.byref(a)) == 18
                K.wgrad(gd, xd, geom, Co, Ci, dw, Ci * 16, 16, db=db, accumulate=True, accumulate_bias=True)      # twice the gradient
                dw, db = dw * 0.5, db * 0.5
            errs[name] = (relerr(dw.cpu().double(), wc.grad), relerr(db.cpu().double(), bc.grad))
        finally:
            L.mtd_conv_wgrad_override(-1, -1)
    assert errs["wino_s2"][0] < max(TOL * 1e-2, 2 * errs["other"][0]) and errs["wino_s2"][1] < max(TOL * 1e-2, 2 * errs["other"][1]), errs


@pytest.mark.parametrize("case", [(2, 64, 64, False), (3, 32, 64, True), (5, 64, 32, False), (32, 64, 64, True), (3, 34, 40, False), (1, 38, 36, True)])
def test_winograd_weight_gradient_32x32_block_vs_float64(hip_lib, case):
    """F(2x2, 3x3) weight gradient of the generator's 32 -> 32 channel layers on one 32 x 32 block (csrc/conv_wgrad_wino32.h, plan 19;
    networks.py:95-164): forward (Conv2d) and transposed (ConvTranspose2d: flipped taps, IOHW view) tap order, bias gradient,
    accumulation; (3, 34, 40) / (1, 38, 36): 20 / 18 tiles per tile row -- chunks of 16 tiles that straddle rows and images, a ragged last chunk --
    against autograd in float64, with the row-window kernel's error on the same operands as the
    yardstick (at most twice its error, and 1e-3 in any case).  case = (B, H, W, transposed)."""
    from mtd_gan_amd import _lib
    from mtd_gan_amd import kernels as K
    B, H, W, transposed = case
    x = rnd(B, 32, H, W, seed=51)
    gy = rnd(B, 32, H, W, seed=52)
    if transposed:
        wc = torch.zeros(32, 32, 3, 3, dtype=torch.double, requires_grad=True)           # ConvTranspose2d weight: [in][out][kh][kw]
        bc = torch.zeros(32, dtype=torch.double, requires_grad=True)
        (F.conv_transpose2d(x.double(), wc, bc, stride=1, padding=1) * gy.double()).sum().backward()
        geom, sn, sc = K.geom_dgrad_s1(B, H, W, 3, 1), 9, 288
    else:
        wc = torch.zeros(32, 32, 3, 3, dtype=torch.double, requires_grad=True)
        bc = torch.zeros(32, dtype=torch.double, requires_grad=True)
        (F.conv2d(x.double(), wc, bc, padding=1) * gy.double()).sum().backward()
        geom, sn, sc = K.geom_fwd(B, H, W, 3, 1, 1), 288, 9
    xd, gd = nhwc(x), nhwc(gy)
    L = _lib.lib()
    errs = {}
    for name, cfg in (("wino32", 19), ("row", 8 if transposed else 7)):
        L.mtd_conv_wgrad_override(cfg, -1)
        try:
            dw = torch.full((32, 32, 3, 3), float("nan"), device="cuda")
            db = torch.full((32,), float("nan"), device="cuda")
            K.wgrad(gd, xd, geom, 32, 32, dw, sn, sc, db=db)
            if name == "wino32":
                a = K.WgradArgs()
                a.g = geom
                a.p, a.p_ld, a.N, a.q, a.q_ld, a.C = gd.data_ptr(), 32, 32, xd.data_ptr(), 32, 32
                a.dw, a.w_sn, a.w_sc = dw.data_ptr(), sn, sc
                assert L.mtd_conv_wgrad_plan_cfg(_lib.C.byref(a)) == 19
                K.wgrad(gd, xd, geom, 32, 32, dw, sn, sc, db=db, accumulate=True, accumulate_bias=True)
                dw, db = dw * 0.5, db * 0.5
            errs[name] = (relerr(dw.cpu().double(), wc.grad), relerr(db.cpu().double(), bc.grad))
        finally:
            L.mtd_conv_wgrad_override(-1, -1)
    assert errs["wino32"][0] < max(TOL * 1e-2, 2 * errs["row"][0]) and errs["wino32"][1] < max(TOL * 1e-2, 2 * errs["row"][1]), errs

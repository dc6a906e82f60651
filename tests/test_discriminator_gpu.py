"""GPU parity of the multi-task discriminator (module surface -> C ABI -> HIP kernels) against the CPU
oracle and the reference-generated golden vectors: eval / train forward (spectral-norm state evolution,
injected dropout masks), input gradient and all 108 parameter gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mtdgan_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


from _metrics import rel  # noqa: E402  (tensor-wide AND element-wise bound)
_rel = rel


def _disc(seed=9):
    from mtd_gan_amd.arch.Ours.networks import Multi_Task_Discriminator_Skip
    dstate = orc.seeded_fill(orc.d_state_shapes(), seed=seed)
    D = Multi_Task_Discriminator_Skip(1, 64)
    assert set(D.state_dict().keys()) == set(orc.d_state_shapes().keys())
    D.load_state_dict(dstate)
    return D.cuda(), dstate


def _masks(n, batch, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(batch, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(n)]


def test_partition_lists_match_reference_order(hip_lib):
    D, _ = _disc()
    names = {id(p): n for n, p in D.named_parameters()}
    assert [names[id(p)] for p in D.shared_parameters()] == orc.d_shared_names()
    assert [names[id(p)] for p in D.task_specific_parameters()] == orc.d_task_specific_names()
    assert [names[id(p)] for p in D.last_shared_parameters()] == ["bconv2.bias", "bconv2.weight_orig"]
    assert sum(p.numel() for p in D.parameters()) == 68432027


def test_eval_forward_vs_golden(hip_lib):
    z = np.load(os.path.join(GOLD, "discriminator.npz"))
    D, _ = _disc()
    D.eval()
    _, y = orc.synthetic_ldct(2, seed=1234)
    with torch.no_grad():
        e, s, r = D(y.cuda())
    assert rel(e, z["eval_enc"]) < TOL and rel(s, z["eval_dec"]) < TOL and rel(r, z["eval_rec"]) < TOL


def test_train_forward_backward_vs_oracle_and_golden(hip_lib):
    z = np.load(os.path.join(GOLD, "discriminator.npz"))
    D, dstate = _disc()
    D.train()
    masks = _masks(2, 2, seed=21)
    D._inject_masks = [m.clone() for m in masks]
    _, y = orc.synthetic_ldct(2, seed=1234)
    yin = y.cuda().requires_grad_(True)
    for it in range(2):                               # two train-mode forwards: u / v evolve
        outs = D(yin)
    assert rel(outs[0].detach(), z["train_enc"]) < TOL
    assert rel(outs[1].detach(), z["train_dec"]) < TOL
    assert rel(outs[2].detach(), z["train_rec"]) < TOL
    sd = D.state_dict()
    assert rel(sd["conv11.weight_u"], z["u_conv11"]) < TOL
    assert rel(sd["bconv2.weight_u"], z["u_bconv2"]) < TOL
    assert rel(sd["c_fc.weight_v"], z["v_c_fc"]) < TOL
    cots = [torch.from_numpy(z[k]) for k in ("cot_enc", "cot_dec", "cot_rec")]
    sum((o * c.cuda()).sum() for o, c in zip(outs, cots)).backward()
    assert rel(yin.grad, z["dinput"]) < TOL
    # float64 oracle as the arbiter for the parameter gradients (see test_generator_gpu.py)
    grads = {}
    for dt in (torch.float32, torch.float64):
        ds = {k: v.to(dt).clone() for k, v in dstate.items()}
        names = orc.d_shared_names() + orc.d_task_specific_names() + ["c_fc.bias", "c_fc.weight_orig"]
        for n in names:
            ds[n] = ds[n].requires_grad_(True)
        leaves = {n: ds[n] for n in names}
        yo = y.to(dt).clone().requires_grad_(True)
        for it in range(2):
            o = orc.discriminator_forward(ds, yo, train=True, drop_mask=masks[it].to(dt))
        sum((a * c.to(dt)).sum() for a, c in zip(o, cots)).backward()
        grads[dt] = {n: leaves[n].grad for n in names}
    worst = 0.0
    for n, p in D.named_parameters():
        ref = grads[torch.float64][n]
        cpu32 = rel(grads[torch.float32][n], ref)
        e = rel(p.grad, ref)
        worst = max(worst, e)
        assert e < max(TOL, 2 * cpu32), (n, e, cpu32)
    for n, gn in zip(z["grad_names"], z["grad_norms"]):
        mine = dict(D.named_parameters())[str(n)].grad.double().norm().item()
        assert abs(mine - gn) <= 5e-3 * gn + 1e-12, (n, mine, gn)


def test_spectral_norm_kernels_vs_torch(hip_lib):
    """mtd_sn_power_iter / mtd_sn_grad on their own against the explicit formulas."""
    from mtd_gan_amd import discriminator_path as DP
    D, dstate = _disc(seed=5)
    P = D._param_dict()
    sig, us, vs = DP._sn_forward(P, True, torch.device("cuda"))
    torch.cuda.synchronize()
    for i, (n, rows, cols) in enumerate(DP.SN_SPECS):
        w = dstate[n + ".weight_orig"].reshape(rows, cols).double()
        u0 = dstate[n + ".weight_u"].double()
        v = torch.nn.functional.normalize(w.t() @ u0, dim=0, eps=1e-12)
        u = torch.nn.functional.normalize(w @ v, dim=0, eps=1e-12)
        sigma = torch.dot(u, w @ v)
        assert rel(P[n + ".weight_v"], v) < 1e-4, n
        assert rel(P[n + ".weight_u"], u) < 1e-4, n
        assert abs(sig[i, 0].item() - sigma.item()) < 1e-4 * abs(sigma.item()), n
        assert abs(sig[i, 1].item() * sigma.item() - 1.0) < 1e-4, n
        assert rel(us[DP.SN_ROW_OFF[i]:DP.SN_ROW_OFF[i] + rows], u) < 1e-4


def test_fused_power_iterations_vs_torch(hip_lib):
    """mtd_sn_power_iter_multi (round 6): the discriminator step's four power iterations with W v of one iteration and W^T (W v) of the
    next in one pass over the weights -- every iteration's saved u, v, sigma and the final in-place state against four sequential
    iterations in float64 (torch.nn.utils.spectral_norm's update, eps 1e-12), and against four calls of mtd_sn_power_iter (same
    values up to rounding: the fused pass normalises after the column sums and adds its row products in another order)."""
    from mtd_gan_amd import discriminator_path as DP
    dev = torch.device("cuda")
    D, dstate = _disc(seed=5)
    P = D._param_dict()
    multi = DP._sn_forward_multi(P, True, dev, 4)
    torch.cuda.synchronize()
    state_multi = {n: (P[n + ".weight_u"].clone(), P[n + ".weight_v"].clone()) for n, _r, _c in DP.SN_SPECS}
    D2, _ = _disc(seed=5)
    P2 = D2._param_dict()
    single = [DP._sn_forward(P2, True, dev) for _ in range(4)]
    torch.cuda.synchronize()
    for i, (n, rows, cols) in enumerate(DP.SN_SPECS):
        w = dstate[n + ".weight_orig"].reshape(rows, cols).double()
        u = dstate[n + ".weight_u"].double()
        ro, co = DP.SN_ROW_OFF[i], DP.SN_COL_OFF[i]
        for it in range(4):
            v = torch.nn.functional.normalize(w.t() @ u, dim=0, eps=1e-12)
            u = torch.nn.functional.normalize(w @ v, dim=0, eps=1e-12)
            sigma = torch.dot(u, w @ v).item()
            sig, us, vs = multi[it]
            assert rel(us[ro:ro + rows], u) < 1e-4 and rel(vs[co:co + cols], v) < 1e-4, (n, it)
            assert abs(sig[i, 0].item() - sigma) < 1e-4 * abs(sigma) and abs(sig[i, 1].item() * sigma - 1.0) < 1e-4, (n, it)
            sig1, us1, vs1 = single[it]
            assert rel(us[ro:ro + rows], us1[ro:ro + rows]) < 2e-5 and rel(vs[co:co + cols], vs1[co:co + cols]) < 2e-5, (n, it)
            assert abs(sig[i, 0].item() - sig1[i, 0].item()) < 1e-5 * abs(sigma), (n, it)
        assert rel(state_multi[n][0], u) < 1e-4 and rel(state_multi[n][1], v) < 1e-4, n
        assert torch.equal(state_multi[n][0], multi[3][1][ro:ro + rows]) and torch.equal(state_multi[n][1], multi[3][2][co:co + cols]), n


@pytest.mark.parametrize("paired", [False, True])
def test_sn_grad_kernel_vs_formula(hip_lib, paired):
    """mtd_sn_grad alone: g_out += G/sigma - <G, W>/sigma^2 u v^T (torch.nn.utils.spectral_norm's backward with u, v
    detached), for one and for two passes over the same weight, float4 and unaligned layers, 1..many blocks."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(77 + paired)
    shapes = [(64, 9), (64, 576), (130, 37), (512, 8192), (96, 1028)]
    structs, keep, want = [], [], []
    for (rows, cols) in shapes:
        def rnd(*s):
            return torch.randn(*s, generator=g)
        W, G1, G2 = rnd(rows, cols), rnd(rows, cols), rnd(rows, cols)
        u1, v1, u2, v2 = rnd(rows), rnd(cols), rnd(rows), rnd(cols)
        s1, s2 = 1.7, 0.6
        out0 = rnd(rows, cols)

        def corr(Gm, u, v, s):
            Gd, Wd = Gm.double(), W.double()
            return Gd / s - (Gd * Wd).sum() / (s * s) * torch.outer(u.double(), v.double())
        ref = out0.double() + corr(G1, u1, v1, s1) + (corr(G2, u2, v2, s2) if paired else 0.0)
        t = {k: x.to(dev).contiguous() for k, x in dict(W=W, G1=G1, G2=G2, u1=u1, v1=v1, u2=u2, v2=v2, out=out0).items()}
        t["s1"] = torch.tensor([s1, 1.0 / s1], device=dev)
        t["s2"] = torch.tensor([s2, 1.0 / s2], device=dev)
        keep.append(t)
        want.append(ref)
        s = _lib.SnGradLayer()
        s.G, s.w, s.u, s.v, s.sigma = (t[k].data_ptr() for k in ("G1", "W", "u1", "v1", "s1"))
        s.g_out, s.rows, s.cols, s.accumulate = t["out"].data_ptr(), rows, cols, 1
        if paired:
            s.G2, s.u2, s.v2, s.sigma2 = (t[k].data_ptr() for k in ("G2", "u2", "v2", "s2"))
        structs.append(s)
    L = _lib.lib()
    tab, host = K.device_table(structs, dev)
    need = L.mtd_sn_grad_ws_bytes(C.cast(host, C.c_void_p), len(structs))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    K.check(L.mtd_sn_grad(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), ws.data_ptr(), K.stream_ptr()), "mtd_sn_grad")
    torch.cuda.synchronize()
    for t, ref, shp in zip(keep, want, shapes):
        err = rel(t["out"], ref)       # fp32 rounding of <G, W> over up to 4 Mi products, relative to the largest entry
        assert err < 2e-5, (shp, err)


@pytest.mark.parametrize("case", [(6, 4, 64, 128, 3, 1, 0.2, True, False), (5, 2, 32, 64, 3, 1, 0.2, False, True), (8, 2, 64, 64, 4, 2, 1.0, True, True),
                                  (16, 1, 128, 256, 1, 1, 0.2, True, False), (4, 4, 512, 512, 3, 1, 0.2, True, True)])
def test_sn_grad_activation_side_dot_vs_weight_side(hip_lib, case):
    """mtd_sn_grad_layer.act_* (round 6): the correction's <G, W> taken from the cotangent of the layer's output and the saved
    activation -- sigma * sum gy (y - b), y = a > 0 ? a : a / slope -- instead of from G and W.  A real layer (torch, float64):
    y = conv(x, W) / sigma + b, a = LeakyReLU(y), G = the weight gradient of <gy, conv(x, W)>; both forms of mtd_sn_grad on the same
    G against the formula.  case = (B, map side of the OUTPUT, Cin, Cout, k, stride, slope, paired, second cotangent)."""
    import ctypes as C
    from mtd_gan_amd import _lib, kernels as K
    B, h, ci, co, k, st, slope, paired, second = case
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1234 + B)
    pad = (k - 1) // 2 if st == 1 else 1
    x = torch.randn(B, ci, h * st, h * st, generator=g, dtype=torch.double)
    W = (torch.randn(co, ci, k, k, generator=g, dtype=torch.double) / (ci * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(co, generator=g, dtype=torch.double) * 0.3
    Bh = B // 2 if paired else B
    sig = [1.7, 0.6]
    lin = F.conv2d(x, W, None, st, pad)
    scale = torch.cat([torch.full((Bh,), 1.0 / sig[0]), torch.full((B - Bh,), 1.0 / sig[1])]).double().view(B, 1, 1, 1)
    y = lin * scale + b.view(1, co, 1, 1)
    a = F.leaky_relu(y, slope) if slope != 1.0 else y
    gy1 = torch.randn(y.shape, generator=g, dtype=torch.double)
    gy2 = torch.randn(y.shape, generator=g, dtype=torch.double) if second else None
    gy = gy1 + (gy2 if second else 0.0)
    Gs = []
    for lo, hi in ((0, Bh), (Bh, B)) if paired else ((0, B),):
        Gs.append(torch.autograd.grad((gy[lo:hi] * lin[lo:hi]).sum(), W, retain_graph=True)[0].detach())
    rows, cols = co, ci * k * k
    Wm = W.detach().reshape(rows, cols)
    u = [torch.randn(rows, generator=g, dtype=torch.double) for _ in range(2)]
    v = [torch.randn(cols, generator=g, dtype=torch.double) for _ in range(2)]
    out0 = torch.randn(rows, cols, generator=g, dtype=torch.double)
    ref = out0.clone()
    for i, Gm in enumerate(Gs):
        Gm = Gm.reshape(rows, cols)
        ref += Gm / sig[i] - (Gm * Wm).sum() / sig[i] ** 2 * torch.outer(u[i], v[i])
    f32 = lambda t: t.float().to(dev).contiguous()
    nhwc = lambda t: t.permute(0, 2, 3, 1).float().contiguous().to(dev)
    t = dict(W=f32(Wm), G1=f32(Gs[0].reshape(rows, cols)), G2=f32(Gs[-1].reshape(rows, cols)), u1=f32(u[0]), v1=f32(v[0]), u2=f32(u[1]), v2=f32(v[1]),
             b=f32(b), gy=nhwc(gy1), a=nhwc(a), s1=torch.tensor([sig[0], 1 / sig[0]], device=dev), s2=torch.tensor([sig[1], 1 / sig[1]], device=dev))
    if second:
        t["gy2"] = nhwc(gy2)
    L = _lib.lib()
    got = {}
    if paired:      # (mtd_wgrad_args.half_scale leaves this in ONE buffer: mtd_sn_grad_layer.prescaled)
        t["Gpre"] = f32(Gs[0].reshape(rows, cols) / sig[0] + Gs[1].reshape(rows, cols) / sig[1])
    for form in ("weights", "activations") + (("prescaled",) if paired else ()):
        out = f32(out0)
        s = _lib.SnGradLayer()
        s.G, s.w, s.u, s.v, s.sigma = (t[n].data_ptr() for n in ("Gpre" if form == "prescaled" else "G1", "W", "u1", "v1", "s1"))
        s.g_out, s.rows, s.cols, s.accumulate = out.data_ptr(), rows, cols, 1
        if paired:
            s.u2, s.v2, s.sigma2 = (t[n].data_ptr() for n in ("u2", "v2", "s2"))
            if form == "prescaled":
                s.prescaled = 1
            else:
                s.G2 = t["G2"].data_ptr()
        if form != "weights":
            M = B * h * h
            assert M < cols
            s.act_gy, s.act_gy_ld, s.act_a, s.act_a_ld, s.act_bias = t["gy"].data_ptr(), co, t["a"].data_ptr(), co, t["b"].data_ptr()
            if second:
                s.act_gy2, s.act_gy2_ld = t["gy2"].data_ptr(), co
            s.act_M, s.act_M_first, s.act_inv_slope = M, Bh * h * h, 1.0 / slope
        tab, host = K.device_table([s], dev)
        ws = torch.empty(L.mtd_sn_grad_ws_bytes(C.cast(host, C.c_void_p), 1), dtype=torch.uint8, device=dev)
        K.check(L.mtd_sn_grad(tab.data_ptr(), C.cast(host, C.c_void_p), 1, ws.data_ptr(), K.stream_ptr()), "mtd_sn_grad")
        torch.cuda.synchronize()
        got[form] = out
        assert rel(out, ref) < 2e-5, (form, rel(out, ref))
    # the two forms agree far inside the bound of either (the only difference is the rounding of one scalar per pass)
    assert rel(got["activations"], got["weights"].double()) < 5e-6


@pytest.mark.parametrize("shape", [(512, 512, 3, 2), (64, 64, 2, 32), (256, 256, 2, 8), (128, 32, 2, 16)])
def test_upsample_block_standalone(hip_lib, shape):
    """arch/Ours/networks.py:166-175 called directly (the discriminator runs the same two steps in place in its decoder
    buffers): UpsampleBlock(2, Cin, Cout) = Conv2d(Cin, 4 Cout, 1) + PixelShuffle(2), forward and all three gradients against
    the same torch modules on the CPU; channel counts outside the kernels' multiples of 32 are refused."""
    import torch.nn as nn
    from mtd_gan_amd.arch.Ours.networks import UpsampleBlock
    cin, cout, B, r = shape
    torch.manual_seed(5)
    blk = UpsampleBlock(2, cin, cout)
    ref = nn.Sequential(nn.Conv2d(cin, cout * 4, 1, 1, 0), nn.PixelShuffle(2))
    ref[0].load_state_dict(blk.upsample[0].state_dict())
    assert list(blk.state_dict().keys()) == ["upsample.0.weight", "upsample.0.bias"]
    x = torch.randn(B, cin, r, r)
    cot = torch.randn(B, cout, 2 * r, 2 * r)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * cot).sum().backward()
    blk.cuda()
    xg = x.cuda().requires_grad_(True)
    yg = blk(xg)
    assert yg.shape == yr.shape
    (yg * cot.cuda()).sum().backward()
    rel = lambda a, b: _rel(a.detach(), b.detach())
    assert rel(yg, yr) < 1e-3
    assert rel(xg.grad, xr.grad) < 1e-3
    assert rel(blk.upsample[0].weight.grad, ref[0].weight.grad) < 1e-3
    assert rel(blk.upsample[0].bias.grad, ref[0].bias.grad) < 1e-3
    with pytest.raises(RuntimeError):
        blk(x)                                           # CPU tensors are refused: no fallback
    with pytest.raises(NotImplementedError):
        UpsampleBlock(2, 16, 8).cuda()(torch.zeros(1, 16, 4, 4, device="cuda"))

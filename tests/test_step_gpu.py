"""GPU parity of the full training iteration (engine.train_MTD_GAN_Ours semantics: D step with PCGrad,
AdamW, G step, AdamW) against the CPU oracle and the reference-generated golden step, plus the loss,
PCGrad and AdamW kernels on their own."""
import json
import math
import os
import random

import pytest
import torch
import torch.nn.functional as F

import mtdgan_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


from _metrics import rel  # noqa: E402  (tensor-wide AND element-wise bound)


def test_loss_kernels_vs_torch(hip_lib):
    from mtd_gan_amd import losses as L
    g = torch.Generator().manual_seed(3)
    a = torch.rand(4, 1, 64, 64, generator=g)
    b = torch.rand(4, 1, 64, 64, generator=g)
    x, y = orc.synthetic_ldct(4, seed=9)
    s = torch.randn(4, 1, generator=g)
    for name, hip, ref in [
        ("ls_gan", lambda t: L.ls_gan(t, 1.0), lambda t: orc.ls_gan(t, 1.0)),
        ("nds", lambda t: L.NDS_Loss(t, 1.0, (x - y).to(t.device)), lambda t: orc.nds_loss(t, 1.0, x - y)),
        ("l1", lambda t: L.l1_loss(t, b.to(t.device)), lambda t: F.l1_loss(t, b)),
        ("mse", lambda t: L.mse_loss(t, b.to(t.device)), lambda t: F.mse_loss(t, b)),
        ("charb", lambda t: L.CharbonnierLoss()(t, b.to(t.device)), lambda t: orc.charbonnier(t, b)),
        ("edge", lambda t: L.EdgeLoss()(t, b.to(t.device)), lambda t: orc.edge_loss(t, b)),
    ]:
        ac = a.clone().requires_grad_(True)
        ad = a.cuda().requires_grad_(True)
        lr_, lh = ref(ac), hip(ad)
        lr_.backward()
        lh.backward()
        assert abs(lh.item() - lr_.item()) < 1e-5 * abs(lr_.item()) + 1e-8, name
        assert rel(ad.grad, ac.grad) < 1e-4, name
    sc, sd = s.clone().requires_grad_(True), s.cuda().requires_grad_(True)
    orc.ls_gan(sc, 0.0).backward()
    L.ls_gan(sd, 0.0).backward()
    assert rel(sd.grad, sc.grad) < 1e-5


def test_pcgrad_kernels_vs_gram_form(hip_lib):
    from mtd_gan_amd import kernels as K
    cases = json.load(open(os.path.join(GOLD, "pcgrad.json")))
    g = torch.Generator().manual_seed(5)
    n = 1_000_003
    for trial in range(4):
        base = torch.randn(3, n, generator=g)
        if trial % 2 == 0:
            base[1] = -0.7 * base[0] + 0.3 * base[1]
        if trial == 3:
            base[2] = -0.2 * base[1] + 0.05 * base[2]
        orders = cases[trial]["orders"]
        vecs = [base[i].cuda().contiguous() for i in range(3)]
        gram = K.pcgrad_gram(vecs)
        ref_gram = (base.double() @ base.double().t())
        assert rel(gram.reshape(3, 3), ref_gram) < 1e-5
        merged = torch.empty(n, device="cuda")
        od = torch.tensor([j for o in orders for j in o], dtype=torch.int32, device="cuda")
        coeff = K.pcgrad_combine(vecs, gram, od, merged)
        w = orc.pcgrad_coefficients(ref_gram.tolist(), orders)
        assert rel(coeff[:3], torch.tensor(w)) < 1e-4
        ref = sum(wk * base[k].double() for k, wk in enumerate(w))
        assert rel(merged, ref) < 1e-4
        random.seed(cases[trial]["seed"])
        lit = orc.pcgrad_merge([base[i].double() for i in range(3)])      # literal reference algorithm, same shuffle stream
        assert rel(merged, lit) < 1e-4


def test_fused_adamw_vs_torch(hip_lib):
    from mtd_gan_amd.optimizers import FusedAdamW
    g = torch.Generator().manual_seed(1)
    shapes = [(64, 32, 3, 3), (513,), (1,), (512, 512)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().cuda().requires_grad_(True) for p in ps]
    o_ref = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    o_mine = FusedAdamW(mine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    for it in range(3):
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad = gr.clone()
            m.grad = gr.cuda()
        if it == 1:
            ref[1].grad = None
            mine[1].grad = None                 # skipped parameter keeps its step count (torch semantics)
        o_ref.step()
        o_mine.step()
    for r, m in zip(ref, mine):
        assert rel(m.detach(), r.detach()) < 1e-6


def _model(batch):
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    z = json.load(open(os.path.join(GOLD, "step_seeded.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"]).items()})
    m = MTD_GAN_Method()
    assert set(m.state_dict().keys()) == set(full.keys()) and len(full) == 326
    m.load_state_dict(full)
    m.cuda().train()
    g = torch.Generator().manual_seed(z["mask_seed"])
    masks = [(torch.rand(batch, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    return m, full, masks, z


def test_full_step_vs_golden_and_oracle(hip_lib):
    from mtd_gan_amd import engine
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    m, full, masks, z = _model(2)
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    x, y = orc.synthetic_ldct(z["batch"], seed=z["data_seed"])
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    oD = FusedAdamW(m.Discriminator.parameters(), lr=z["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oG = FusedAdamW(m.Generator.parameters(), lr=z["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    random.seed(77)                                    # the seed the golden step was generated with
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 2, wm)
    # ---- the 17 logged scalars against the reference's own output
    for k, v in z["stats"].items():
        assert abs(stats[k] - v) <= TOL * abs(v) + 2e-6, (k, stats[k], v)
    # ---- updated parameters / spectral-norm state against the reference's post-step state_dict samples
    post = m.state_dict()
    # an AdamW step moves every weight by ~lr regardless of gradient scale, so compare the *updates*
    bad = []
    for k, samples in z["post_samples"].items():
        t = post[k].reshape(-1).cpu()
        t0 = full[k].reshape(-1)
        nel = t.numel()
        for i, s in enumerate(samples):
            idx = (i * 2654435761 + 12345) % nel
            if k.endswith(("weight_u", "weight_v")):
                ok = abs(t[idx].item() - s) <= TOL * max(abs(s), 1e-3)
            else:
                ok = abs((t[idx].item() - t0[idx].item()) - (s - t0[idx].item())) <= 0.05 * z["lr"] + 1e-9
            if not ok:
                bad.append((k, i, t[idx].item(), s))
    assert len(bad) <= 0.01 * 4 * len(z["post_samples"]), bad[:10]      # sign flips of ~0 gradients are the only tolerated misses
    assert torch.equal(post["Discriminator.c_fc.weight_orig"].cpu(), full["Discriminator.c_fc.weight_orig"])   # frozen (quirk 1)
    # ---- SURVEY 8d config-3 acceptance: PSNR of the generator AFTER its update against the reference's (metrics.py:184-197)
    z32 = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    _post_step_psnr(m, x, y, z32["b2_post_metrics"], z32["b2_post_pred_sample"], 8)


def _post_step_psnr(m, x, y, want, pred_sample, stride):
    from mtd_gan_amd import metrics as M
    with torch.no_grad():
        pred = m.Generator(x.cuda())
    got = M.compute_PSNR(x.cuda(), y.cuda(), pred.clip(0, 1))[2]
    assert abs(float(got) - want["psnr"]) <= 0.01, (float(got), want["psnr"])                  # north_star: within 0.01 dB
    assert abs(float(M.compute_RMSE(x.cuda(), y.cuda(), pred.clip(0, 1))[2]) - want["rmse"]) <= TOL * want["rmse"]
    assert rel(pred[:, 0, ::stride, ::stride], torch.tensor(pred_sample)) < TOL


@pytest.mark.parametrize("pipe", ["fp32", "bf16x3"])
def test_full_step_b32_vs_golden(hip_lib, pipe):
    """(pipe "bf16x3", round 5: the same iteration with mtd_set_option("wino_split", 1) -- the 3x3 stride-1 trunk layers' Winograd
    products as six bf16 MFMA products of exact three-way splits -- held to the SAME golden values and bounds.)
    BASELINE configs[2] at its full size: one iteration on 32 patches from the seeded-fill state (O(1) activations, all
    three task gradients non-degenerate) against the step the reference itself ran -- 17 logged scalars, sampled
    post-step parameters and spectral-norm vectors, the Gram matrix of the task gradients, the generator's PSNR after
    the step.  The D optimizer has the reference's two-group layout (train.py:122-124: second group empty, lr 0.025)."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    from mtd_gan_amd import _lib, kernels as K

    def set_split(on):
        assert _lib.lib().mtd_set_option(b"wino_split", int(on)) == 0
        K._wino_px_cache.clear()
        K._igemm_ws_cache.clear()
        K.weights_changed(None)
    set_split(pipe == "bf16x3")
    try:
        _full_step_b32(engine, WeightMethods, FusedAdamW)
    finally:
        set_split(False)


def _full_step_b32(engine, WeightMethods, FusedAdamW):
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    m, full, masks, _ = _model(32)
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    x, y = orc.synthetic_ldct(32, seed=z["data_seed"])
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = FusedAdamW([dict(params=m.Discriminator.parameters(), lr=z["lr"], **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = FusedAdamW(m.Generator.parameters(), lr=z["lr"], **kw)
    assert len(oD.param_groups) == 2 and oD.param_groups[1]["params"] == []
    random.seed(77)
    captured = {}
    real_d_loss = m.d_loss

    def d_loss(a, b):
        out = real_d_loss(a, b)
        captured["tape"] = out[0]._mtd_tape
        return out
    m.d_loss = d_loss
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 32, wm)
    for k, v in z["stats"].items():
        assert abs(stats[k] - v) <= TOL * abs(v) + 2e-6, (k, stats[k], v)
    tape = captured["tape"]
    assert tape.orders == z["orders"]
    gram_ref = torch.tensor(z["gram"], dtype=torch.float64)
    scale = torch.sqrt(torch.outer(gram_ref.diag(), gram_ref.diag()))
    assert ((tape.gram.reshape(3, 3).cpu() - gram_ref).abs() / scale).max().item() < TOL
    assert rel(tape.coeff[:3], torch.tensor(z["pc_weights"])) < TOL
    post = m.state_dict()
    bad = []
    for k, samples in z["post_samples"].items():
        t, t0 = post[k].reshape(-1).cpu(), full[k].reshape(-1)
        for i, sv in enumerate(samples):
            idx = (i * 2654435761 + 12345) % t.numel()
            if k.endswith(("weight_u", "weight_v")):
                ok = abs(t[idx].item() - sv) <= TOL * max(abs(sv), 1e-3)
            else:             # a first AdamW step moves a weight by lr * sign(gradient): compare the updates
                ok = abs((t[idx].item() - t0[idx].item()) - (sv - t0[idx].item())) <= 0.05 * z["lr"] + 1e-9
            if not ok:
                bad.append((k, i, t[idx].item(), sv))
    assert len(bad) <= 0.01 * 4 * len(z["post_samples"]), bad[:10]
    assert torch.equal(post["Discriminator.c_fc.weight_orig"].cpu(), full["Discriminator.c_fc.weight_orig"])
    _post_step_psnr(m, x, y, z["post_metrics"], z["post_pred_sample"], 16)


@pytest.mark.parametrize("pipe", ["fp32", "bf16x3"])
def test_g_step_gradients_b32_vs_reference_samples(hip_lib, pipe):
    """(pipe "bf16x3": with mtd_set_option("wino_split", 1) -- the discriminator's 3x3 stride-1 layers on the split-bf16 Winograd
    kernel, through which the G step's data gradient flows -- under the same per-tensor bounds.)
    The generator's gradients of the G step at BASELINE size (engine.py:50-52: g_loss.backward() after the discriminator's
    update; networks.py:1994-2009) against the reference's own: per parameter 6 sampled ELEMENTS and the tensor's norm
    (tests/golden/g_grad_samples_b32.json, written by oracle/pin_grad_samples.py from a run of the reference's engine loop),
    and the norms committed with the golden step (step_seeded_b32.json `g_grad_norms`).  Bound per tensor: max(1e-3, 2 x the
    reference's own fp32-vs-float64 error on that tensor) of its max-abs, float64 as the arbiter."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    from mtd_gan_amd import _lib, kernels as K

    def set_split(on):
        assert _lib.lib().mtd_set_option(b"wino_split", int(on)) == 0
        K._wino_px_cache.clear()
        K._igemm_ws_cache.clear()
        K.weights_changed(None)
    set_split(pipe == "bf16x3")
    try:
        _g_step_gradients_b32(engine, WeightMethods, FusedAdamW)
    finally:
        set_split(False)


def _g_step_gradients_b32(engine, WeightMethods, FusedAdamW):
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    gs = json.load(open(os.path.join(GOLD, "g_grad_samples_b32.json")))
    m, full, masks, _ = _model(32)
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    x, y = orc.synthetic_ldct(32, seed=z["data_seed"])
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = FusedAdamW([dict(params=m.Discriminator.parameters(), lr=z["lr"], **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = FusedAdamW(m.Generator.parameters(), lr=z["lr"], **kw)
    random.seed(77)
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 32, wm)
    assert abs(stats["g_loss"] - gs["g_loss_ref32"]) <= TOL * abs(gs["g_loss_ref32"])
    names = [n for n, _ in m.Generator.named_parameters()]
    assert ["Generator." + n for n in names] == list(gs["grads"].keys()) and len(names) == 128
    bad, worst, nel = [], 0.0, 0
    for n, p in m.Generator.named_parameters():
        e = gs["grads"]["Generator." + n]
        assert p.grad is not None, n                                  # (optimizer.step() leaves .grad in place, as torch.optim does)
        g = p.grad.detach().reshape(-1).double().cpu()
        bound = max(TOL, 2 * e["err32"])
        for i, want in enumerate(e["f64"]):
            idx = (i * 2654435761 + 12345) % g.numel()
            err = abs(g[idx].item() - want) / (e["maxabs"] + 1e-30)
            worst = max(worst, err / bound)
            nel += 1
            if err > bound:
                bad.append((n, i, g[idx].item(), want, err, bound))
        # the tensor's norm: against the float64 value, and against the norm committed with the golden step (the reference's fp32)
        nb = max(TOL, 2 * abs(e["norm_ref32"] - e["norm_f64"]) / (e["norm_f64"] + 1e-30))
        got = g.norm().item()
        if abs(got - e["norm_f64"]) > nb * e["norm_f64"] or abs(got - z["g_grad_norms"]["Generator." + n]) > 2 * nb * e["norm_f64"]:
            bad.append((n, "norm", got, e["norm_f64"], z["g_grad_norms"]["Generator." + n]))
    assert not bad, (len(bad), bad[:8])
    assert nel == 6 * 128
    print(f"G-step gradients at B = 32: {nel} elements + 128 norms, worst error / bound {worst:.3f}")


def test_config1_anchor_on_the_hip_path(hip_lib):
    """BASELINE configs[0] (`MTD_GAN_Method`, --batch-size 2 => 16 patches, 1 train step; a CPU run in the reference): the
    same run on the HIP path.  The reference's default-init state (seeds 2024) is rebuilt from the recorded draw tape and
    verified against per-tensor checksums, the inputs continue the same generator stream, the five dropout masks the
    reference drew are injected, `random` is seeded as the reference seeded it; the 17 logged scalars must match the
    reference's (tests/golden/config1_replay.json == config1_anchor.json == BASELINE.md section 3)."""
    import replay_config1
    from mtd_gan_amd import engine
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    z, state, x, y, masks = replay_config1.load()
    anchor = json.load(open(os.path.join(GOLD, "config1_anchor.json")))
    assert anchor["stats"] == z["stats"]
    m = MTD_GAN_Method()
    m.load_state_dict(state)
    m.cuda().train()
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    dev = torch.device("cuda")
    wm = WeightMethods(method="pcgrad", n_tasks=3, device=dev)
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = FusedAdamW([dict(params=m.Discriminator.parameters(), lr=z["lr"], **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = FusedAdamW(m.Generator.parameters(), lr=z["lr"], **kw)
    random.seed(z["seed"])
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, dev, 0, 0, z["batch_size_arg"], wm)
    assert list(stats.keys()) == list(z["stats"].keys())                       # same 17 keys, same order
    for k, v in z["stats"].items():
        assert abs(stats[k] - v) <= TOL * abs(v) + 2e-6, (k, stats[k], v)
    for k, v in anchor["known"].items():                                       # BASELINE.md's known answer
        assert abs(stats[k] - v) <= TOL * abs(v), (k, stats[k], v)
    assert D_frozen(m, state)


def D_frozen(m, state):
    post = m.state_dict()
    return torch.equal(post["Discriminator.c_fc.weight_orig"].cpu(), state["Discriminator.c_fc.weight_orig"])


def test_pcgrad_optimizer_wrapper_vs_reference(hip_lib):
    """module/pcgrad.py::PCGrad on the HIP Gram / coefficient / axpy kernels against the gradients the
    reference class produced on its own self-test nets (pcgrad.py:144-195) and on a three-objective conflict case."""
    import torch.nn as nn
    from mtd_gan_amd.module.pcgrad import PCGrad
    from mtd_gan_amd.optimizers import FusedAdamW
    cases = json.load(open(os.path.join(GOLD, "pcgrad_wrapper.json")))

    class Net(nn.Module):
        def __init__(self, heads):
            super().__init__()
            self._linear = nn.Linear(3, 2 if heads else 4)
            if heads:
                self._head1, self._head2 = nn.Linear(2, 4), nn.Linear(2, 4)

        def forward(self, t):
            f = self._linear(t)
            return (self._head1(f), self._head2(f)) if hasattr(self, "_head1") else f
    for c in cases:
        net = Net(c["tag"] != "TestNet")
        net.load_state_dict({k: torch.tensor(v) for k, v in c["state"].items()})
        net.cuda()
        x, ys = torch.tensor(c["x"]).cuda(), [torch.tensor(v).cuda() for v in c["ys"]]
        pc = PCGrad(FusedAdamW(net.parameters(), lr=1e-3))
        pc.zero_grad()
        if c["tag"] == "TestNet":
            objs = [nn.L1Loss()(net(x), ys[0]), nn.MSELoss()(net(x), ys[0])]
        elif len(ys) == 1:
            objs = [nn.MSELoss()(net(x)[0], ys[0]), nn.MSELoss()(net(x)[1], ys[0])]
        else:
            objs = [nn.MSELoss()(net(x)[0], ys[0]), nn.MSELoss()(net(x)[0], -ys[0] + 0.1), nn.L1Loss()(net(x)[1], ys[1])]
        random.seed(c["rseed"])
        pc.pc_backward(objs)
        for n, p in net.named_parameters():
            assert p.grad is not None and rel(p.grad, torch.tensor(c["grads"][n])) < 1e-4, (c["tag"], n)
        before = [p.detach().clone() for p in net.parameters()]
        pc.step()                                                       # the wrapped optimizer consumes the written gradients
        assert all(not torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
        pc.zero_grad()
        assert all(p.grad is None for p in net.parameters())
    with pytest.raises(ValueError):
        PCGrad(FusedAdamW(net.parameters(), lr=1e-3), reduction="").pc_backward([nn.MSELoss()(net(x)[0], ys[0])])


def test_d_step_task_gradients_vs_oracle(hip_lib):
    """The three per-task shared gradients (what PCGrad projects), their Gram matrix, the merged gradient and the
    task-specific gradients against the float64 oracle, PER TENSOR (40 shared tensors x 3 tasks, 66 task-specific ones):
    error relative to that tensor's own max-abs, bound max(1e-3, 2 x the error of the reference's own fp32 CPU
    arithmetic against float64 on the same tensor) -- the criterion of DESIGN.md section 4, applied where it is earned."""
    from mtd_gan_amd.module.weight_methods import WeightMethods
    m, full, masks, z = _model(2)
    m.Discriminator._inject_masks = [k.clone() for k in masks[:4]]
    x, y = orc.synthetic_ldct(2, seed=1234)
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    random.seed(77)
    losses, details = m.d_loss(x.cuda(), y.cuda())
    D = m.Discriminator
    wm.backward(losses=losses, shared_parameters=list(D.shared_parameters()), task_specific_parameters=list(D.task_specific_parameters()),
                last_shared_parameters=list(D.last_shared_parameters()))
    tape = losses._mtd_tape
    shared = ["Discriminator." + n for n in orc.d_shared_names()]
    tspec = ["Discriminator." + n for n in orc.d_task_specific_names()]

    def oracle(dtype):
        st = {k: v.to(dtype).clone() for k, v in full.items()}
        for n in shared + tspec:
            st[n] = st[n].requires_grad_(True)
        lo, _, _ = orc.d_loss(st, x.to(dtype), y.to(dtype), [k.to(dtype) for k in masks[:4]])
        sp = [st[n] for n in shared]
        per_task = [torch.autograd.grad(lo[i], sp, retain_graph=True) for i in range(3)]
        ts = torch.autograd.grad(lo.sum(), [st[n] for n in tspec])
        return lo.detach(), per_task, ts
    lo, g64, ts64 = oracle(torch.float64)
    _, g32, ts32 = oracle(torch.float32)
    assert rel(losses, lo) < TOL
    sizes = [g.numel() for g in g64[0]]
    bad, worst = [], 0.0

    def check(tag, hip, r64, r32):
        nonlocal worst
        den = r64.abs().max().item() + 1e-30
        e_hip = (hip.double().cpu() - r64).abs().max().item() / den
        e_cpu = (r32.double() - r64).abs().max().item() / den
        worst = max(worst, e_hip)
        if e_hip > max(TOL, 2 * e_cpu):
            bad.append((tag, f"{e_hip:.2e}", f"cpu32 {e_cpu:.2e}"))
    for i in range(3):
        ofs = 0
        for n, sz, r64, r32 in zip(shared, sizes, g64[i], g32[i]):
            check(f"task{i} {n}", tape.task_vectors[i][ofs:ofs + sz].reshape(r64.shape), r64, r32)
            ofs += sz
    names = {id(p): n for n, p in D.named_parameters()}
    for p, r64, r32 in zip(D.task_specific_parameters(), ts64, ts32):
        check("task-specific " + names[id(p)], p.grad, r64, r32)
    flat = [torch.cat([g.reshape(-1) for g in tg]) for tg in g64]
    gram_ref = torch.stack([torch.stack([torch.dot(a, b) for b in flat]) for a in flat])
    scale = torch.sqrt(torch.outer(gram_ref.diag(), gram_ref.diag()))
    gram_err = ((tape.gram.reshape(3, 3).cpu() - gram_ref).abs() / scale).max().item()
    assert tape.orders == z["orders"]
    w = orc.pcgrad_coefficients(gram_ref.tolist(), tape.orders)
    merged_ref = sum(wk * f for wk, f in zip(w, flat))
    merged32 = sum(wk * torch.cat([g.reshape(-1) for g in tg]).double() for wk, tg in zip(w, g32))
    ofs = 0
    for n, sz in zip(shared, sizes):
        check("merged " + n, tape.task_vectors[3][ofs:ofs + sz], merged_ref[ofs:ofs + sz], merged32[ofs:ofs + sz])
        ofs += sz
    print(f"per-tensor worst rel err vs fp64 oracle {worst:.2e}; Gram err relative to |g_a||g_b| {gram_err:.2e}; {len(bad)} over the bound")
    assert not bad, bad[:12]
    assert gram_err < TOL                                             # relative to |g_a||g_b|
    assert D.c_fc.weight_orig.grad is None


def test_first_pass_overwrites_task_vectors_same_bits(hip_lib, monkeypatch):
    """train_step.DStepTape.run_pcgrad: the first backward pass into a task vector OVERWRITES its spectral-norm weight
    gradients (no zero fill of the 3 x 114 MB, no read of it by the correction kernel).  With the vectors poisoned with NaN
    beforehand, every element must have been written, and the three task vectors, the merged gradient and the task-specific
    gradients must equal the zero-fill-and-accumulate form bit for bit (0 + x == x)."""
    from mtd_gan_amd import train_step as TS
    from mtd_gan_amd.module.weight_methods import WeightMethods
    res = {}
    for mode in ("accumulate", "overwrite"):
        monkeypatch.setattr(TS, "FIRST_WRITE", mode == "overwrite")
        monkeypatch.setattr(TS, "POISON", mode == "overwrite")
        m, full, masks, z = _model(2)
        m.Discriminator._inject_masks = [k.clone() for k in masks[:4]]
        x, y = orc.synthetic_ldct(2, seed=1234)
        wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
        random.seed(77)
        losses, _ = m.d_loss(x.cuda(), y.cuda())
        D = m.Discriminator
        wm.backward(losses=losses, shared_parameters=list(D.shared_parameters()), task_specific_parameters=list(D.task_specific_parameters()),
                    last_shared_parameters=list(D.last_shared_parameters()))
        tape = losses._mtd_tape
        torch.cuda.synchronize()
        res[mode] = (tape.task_vectors.clone(), [p.grad.clone() for p in D.task_specific_parameters()])
    S0, ts0 = res["accumulate"]
    S1, ts1 = res["overwrite"]
    assert torch.isfinite(S1).all()
    assert torch.equal(S0, S1)
    assert all(torch.equal(a, b) for a, b in zip(ts0, ts1))


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [2, 32])
def test_paired_launches_of_the_d_step_same_results(hip_lib, monkeypatch, batch):
    """Round 6: (a) the two decoders' mirror convs as pairs (discriminator_path.PAIR_DECODERS: forward pass over tape 1+2 and the second
    consistency pass), (b) the adversarial and the first consistency pass advanced together with the data gradients of the same layer
    in one launch (LOCKSTEP: disc_backward_lockstep).  Both only change HOW MANY launches carry the same products (a pair launch plans
    its split of K for the whole grid: another grouping of the K sums, 1e-6 per layer): the losses, the three task vectors and the
    task-specific gradients against the one-launch-per-layer, one-pass-after-the-other schedule inside the parity bound (1e-3), at 2
    patches and at the BASELINE's 32; and the schedule is repeatable bit for bit."""
    from mtd_gan_amd import discriminator_path as DPm
    from mtd_gan_amd.module.weight_methods import WeightMethods
    res = {}
    for mode in ("plain", "pairs", "pairs+lockstep", "pairs+lockstep again", "pairs+three passes", "pairs+two groups"):
        monkeypatch.setattr(DPm, "PAIR_DECODERS", mode != "plain")
        monkeypatch.setattr(DPm, "LOCKSTEP", 1 if mode.startswith("pairs+lockstep") else {"pairs+three passes": 3, "pairs+two groups": 2}.get(mode, 0))
        m, full, masks, z = _model(batch)
        m.Discriminator._inject_masks = [k.clone() for k in masks[:4]]
        x, y = orc.synthetic_ldct(batch, seed=1234)
        wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
        random.seed(77)
        losses, _ = m.d_loss(x.cuda(), y.cuda())
        D = m.Discriminator
        wm.backward(losses=losses, shared_parameters=list(D.shared_parameters()), task_specific_parameters=list(D.task_specific_parameters()),
                    last_shared_parameters=list(D.last_shared_parameters()))
        tape = losses._mtd_tape
        torch.cuda.synchronize()
        res[mode] = (losses.detach().clone(), tape.task_vectors.clone(), [p.grad.clone() for p in D.task_specific_parameters()])
    l0, S0, ts0 = res["plain"]
    for mode in ("pairs", "pairs+lockstep", "pairs+three passes", "pairs+two groups"):      # (three passes / two groups: lab options)
        l1, S1, ts1 = res[mode]
        # (the forward pass's pairs move the losses by rounding; the gradients of the deep ReLU stacks amplify a layer's 1e-6 the way
        # they amplify every fp32 rounding difference -- the reference's own fp32 path is 5e-3 from its float64 evaluation on the small
        # deep-layer gradients: measured here 1.8e-4 under the element-wise measure at 32 patches, the parity bound is 1e-3)
        assert rel(l1, l0) < 1e-5, (mode, rel(l1, l0))
        for i in range(3):                      # each task vector against its own scale
            assert rel(S1[i], S0[i]) < 1e-3, (mode, i, rel(S1[i], S0[i]))
        # the task-specific gradients as ONE vector against its largest element (they are views of one bucket; some of its tensors have
        # one to nine elements, each a cancelling sum over 10^5 pixels: their own relative error says nothing), and per tensor loosely
        f0, f1 = torch.cat([t.reshape(-1) for t in ts0]).double(), torch.cat([t.reshape(-1) for t in ts1]).double()
        assert (f1 - f0).abs().max().item() < 1e-4 * f0.abs().max().item(), (mode, (f1 - f0).abs().max().item(), f0.abs().max().item())
        for a, b in zip(ts1, ts0):
            assert (a.double() - b.double()).abs().max().item() <= 2e-2 * b.abs().max().item() + 1e-7 * f0.abs().max().item(), (mode, tuple(b.shape))
    la, Sa, tsa = res["pairs+lockstep"]
    lb, Sb, tsb = res["pairs+lockstep again"]
    assert torch.equal(la, lb) and torch.equal(Sa, Sb) and all(torch.equal(a, b) for a, b in zip(tsa, tsb))


@pytest.mark.gpu
def test_d_step_task_gradients_b32_vs_reference_samples(hip_lib):
    """The D step at BASELINE size (32 patches): the three per-task shared gradients, the merged gradient PCGrad writes and the
    task-specific gradients, PER TENSOR, against tests/golden/grad_samples_b32.json -- elements of the gradients the REFERENCE
    computed at B = 32 (captured on their way into weight_methods.py:449's projection; oracle/pin_grad_samples.py), compared at
    the float64 value with the bound max(1e-3, 2 x the reference's own fp32-vs-float64 error on that tensor) relative to the
    tensor's max-abs.  The post-step checks of test_full_step_b32_vs_golden see update SIGNS; this sees magnitudes."""
    from mtd_gan_amd.module.weight_methods import WeightMethods
    gs = json.load(open(os.path.join(GOLD, "grad_samples_b32.json")))
    m, full, masks, _ = _model(32)
    m.Discriminator._inject_masks = [k.clone() for k in masks[:4]]
    x, y = orc.synthetic_ldct(32, seed=1234)
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    random.seed(77)
    losses, _ = m.d_loss(x.cuda(), y.cuda())
    D = m.Discriminator
    wm.backward(losses=losses, shared_parameters=list(D.shared_parameters()), task_specific_parameters=list(D.task_specific_parameters()),
                last_shared_parameters=list(D.last_shared_parameters()))
    tape = losses._mtd_tape
    assert tape.orders == gs["orders"]
    assert rel(losses, torch.tensor(gs["losses"])) < TOL
    assert rel(tape.coeff[:3], torch.tensor(gs["pc_weights_f64"])) < TOL
    names = {id(p): n for n, p in D.named_parameters()}
    shared = [names[id(p)] for p in D.shared_parameters()]
    assert shared == gs["shared"] and [names[id(p)] for p in D.task_specific_parameters()] == gs["task_specific"]
    bad, worst, checked = [], 0.0, 0

    def check(tag, flat, e):
        nonlocal worst, checked
        idx = [(i * 2654435761 + 12345) % flat.numel() for i in range(len(e["f64"]))]
        got = flat.reshape(-1)[torch.tensor(idx, device=flat.device)].double().cpu().tolist()
        bound = max(TOL, 2 * e["err32"]) * e["maxabs"]
        for j, (a, b) in enumerate(zip(got, e["f64"])):
            worst = max(worst, abs(a - b) / (e["maxabs"] + 1e-30))
            checked += 1
            if abs(a - b) > bound + 1e-30:
                bad.append((tag, j, a, b, bound))
        mx = flat.abs().max().item()
        if abs(mx - e["maxabs"]) > 5e-3 * e["maxabs"] + 1e-30:
            bad.append((tag, "maxabs", mx, e["maxabs"]))
    sizes = [p.numel() for p in D.shared_parameters()]
    for i in range(3):
        ofs = 0
        for n, sz in zip(shared, sizes):
            check(f"task{i} {n}", tape.task_vectors[i][ofs:ofs + sz], gs["tasks"][i][n])
            ofs += sz
    ofs = 0
    for n, sz in zip(shared, sizes):
        check("merged " + n, tape.task_vectors[3][ofs:ofs + sz], gs["merged"][n])
        ofs += sz
    for p in D.task_specific_parameters():
        check("task-specific " + names[id(p)], p.grad, gs["ts"][names[id(p)]])
    print(f"{checked} gradient elements at B=32, worst error relative to the tensor's max-abs {worst:.2e}; {len(bad)} over the bound")
    assert not bad, bad[:12]


def test_readme_batch_of_160_patches(hip_lib):
    """The reference's documented training configuration: `--batch-size 20` (README.md:77) x 8 random crops per slice
    (create_datasets/Mayo.py:126, list_data_collate) = 160 patches per iteration; the paired discriminator passes then run on
    320 images (largest activation 320 x 64 x 64 x 128 floats = 671 MB: still inside the kernels' 32-bit byte offsets).
    One complete iteration from the seeded-fill state must be finite and move every listed parameter; the generator on the
    160-patch batch must equal its 32-patch chunks (patches are independent); the discriminator's three outputs on two of the
    160 patches (eval mode, first and last) must match the CPU oracle."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    B = 160
    m, full, masks, z = _model(B)
    x, y = orc.synthetic_ldct(B, seed=4321)
    xd, yd = x.cuda(), y.cuda()
    # ---- forward properties at this size, before the step
    m.eval()
    with torch.no_grad():
        out = m.Generator(xd)
        for i in range(0, B, 32):
            assert rel(m.Generator(xd[i:i + 32]), out[i:i + 32]) < 1e-6, i
        e, s, r = m.Discriminator(yd)
    dstate = {k[len("Discriminator."):]: v for k, v in full.items() if k.startswith("Discriminator.")}
    pick = torch.tensor([0, B - 1])
    eo, so, ro = orc.discriminator_forward({k: v.clone() for k, v in dstate.items()}, y[pick], train=False)
    assert rel(e[pick.cuda()], eo) < TOL and rel(s[pick.cuda()], so) < TOL and rel(r[pick.cuda()], ro) < TOL
    with torch.no_grad():
        gref = orc.generator_forward({k[len("Generator."):]: v for k, v in full.items() if k.startswith("Generator.")}, x[pick])
    assert rel(out[pick.cuda()], gref) < TOL
    # ---- one iteration
    m.train()
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = FusedAdamW([dict(params=m.Discriminator.parameters(), lr=1e-4, **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, **kw)
    random.seed(77)
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 20, wm)
    assert len(stats) == 17 and all(math.isfinite(v) for v in stats.values()), stats
    post = m.state_dict()
    names = {id(p): n for n, p in m.Discriminator.named_parameters()}
    listed = {"Discriminator." + names[id(p)] for p in list(m.Discriminator.shared_parameters()) + list(m.Discriminator.task_specific_parameters())}
    listed |= {"Generator." + n for n, _ in m.Generator.named_parameters()}
    for k in listed:
        assert torch.isfinite(post[k]).all(), k
        assert not torch.equal(post[k].cpu(), full[k]), k                    # AdamW moved it
    assert torch.equal(post["Discriminator.c_fc.weight_orig"].cpu(), full["Discriminator.c_fc.weight_orig"])


def test_full_batch_step_runs_and_is_finite(hip_lib):
    """BASELINE size (32 patches): one complete iteration; losses finite, every listed parameter gets a
    finite gradient / update, spectral-norm vectors stay unit length."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    torch.manual_seed(0)
    m = MTD_GAN_Method().cuda().train()
    x, y = orc.synthetic_ldct(32, seed=1234)
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
    oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 4, wm)
    assert len(stats) == 17 and all(v == v and abs(v) < 1e6 for v in stats.values())
    after = m.state_dict()
    moved = sum(int(not torch.equal(before[k], after[k])) for k in before)
    assert moved >= 320, moved                       # everything but c_fc.{weight_orig,bias} (frozen) moves
    for k, v in after.items():
        assert torch.isfinite(v).all(), k
        if k.endswith("weight_u") or k.endswith("weight_v"):
            assert abs(v.norm().item() - 1.0) < 1e-4, k


@pytest.mark.parametrize("opt_kind", ["fused", "torch"])
def test_three_iterations_vs_reference(hip_lib, opt_kind):
    """Iterations 2 and 3 pinned to the reference (tests/golden/step_seeded_x3.json, oracle/pin_three_iterations.py: the
    reference's engine.train_MTD_GAN_Ours on three batches in a row from the seeded fill).  What a one-iteration golden step
    cannot see: AdamW's bias correction at steps 2 and 3, the spectral-norm vectors after 15 discriminator forwards, and the
    derived weight views (packed, Winograd-transformed, transposed) that have to be rebuilt after every update -- with
    FusedAdamW (raw-pointer updates, kernels.weights_changed) and with torch.optim.AdamW (in-place updates, version counters).
    Per iteration: the 17 logged scalars at 1e-3; sampled post-iteration parameters (updates within 0.05 lr, <= 1 % of the
    samples may miss, the rule of the one-step tests) and u / v at 1e-3."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    z = json.load(open(os.path.join(GOLD, "step_seeded_x3.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"]).items()})
    m = MTD_GAN_Method()
    m.load_state_dict(full)
    m.cuda().train()
    g = torch.Generator().manual_seed(z["mask_seed"])
    m.Discriminator._inject_masks = [(torch.rand(z["batch"], 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5 * z["iters"])]
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
    kw = dict(lr=z["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    Opt = FusedAdamW if opt_kind == "fused" else torch.optim.AdamW
    oD, oG = Opt(m.Discriminator.parameters(), **kw), Opt(m.Generator.parameters(), **kw)
    random.seed(z["random_seed"])
    for i in range(z["iters"]):
        x, y = orc.synthetic_ldct(z["batch"], seed=z["data_seed"] + i)
        stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), i, 0, z["batch"], wm)
        for k, v in z["per_iter"][i].items():
            assert abs(stats[k] - v) <= TOL * abs(v) + 2e-6, (opt_kind, i, k, stats[k], v)
        post = m.state_dict()
        bad, n = [], 0
        for k, samples in z["post_samples"][i].items():
            t = post[k].reshape(-1).cpu()
            for j, sv in enumerate(samples):
                idx = (j * 2654435761 + 12345) % t.numel()
                tol = TOL * max(abs(sv), 1e-3) if k.endswith(("weight_u", "weight_v")) else 0.05 * z["lr"] + 1e-9
                if abs(t[idx].item() - sv) > tol:
                    bad.append((k, j, t[idx].item(), sv))
                n += 1
        assert len(bad) <= 0.01 * n, (opt_kind, i, len(bad), bad[:8])
    assert torch.equal(m.state_dict()["Discriminator.c_fc.weight_orig"].cpu(), full["Discriminator.c_fc.weight_orig"])      # frozen (quirk 1)
    assert not m.Discriminator._inject_masks                    # 15 masks, 15 discriminator forwards


def test_graph_replay_equals_eager(hip_lib):
    """hipGraph capture/replay of the whole iteration (GraphedTrainStep) must reproduce eager execution:
    same kernels in the same order, host-side state (PCGrad order, AdamW step scalars) refreshed per replay."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    from mtd_gan_amd.train_step import GraphedTrainStep
    x, y = orc.synthetic_ldct(4, seed=77)
    x, y = x.cuda(), y.cuda()
    results = []
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        m = MTD_GAN_Method().cuda().train()
        m.Discriminator.c_drop.p = 0.0                  # no dropout: the two runs draw no random numbers on the device
        wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cuda"))
        oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
        oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
        random.seed(123)
        if mode == "eager":
            for _ in range(5):
                names, vals = engine.train_iteration(m, x, y, oG, oD, wm, None)
        else:
            gs = GraphedTrainStep(m, oG, oD, wm, x, y, warmup=3)       # 3 eager steps, then capture
            for _ in range(2):
                names, vals = gs.step()
        torch.cuda.synchronize()
        results.append(({k: v.clone() for k, v in m.state_dict().items()}, vals.clone(), oD.state[m.Discriminator.enc_out.weight]["step"]))
    (sd_e, v_e, st_e), (sd_g, v_g, st_g) = results
    assert st_e == st_g == 5
    assert rel(v_g, v_e) < 1e-5
    for k in sd_e:
        assert rel(sd_g[k], sd_e[k]) < 1e-5, k


def test_launch_list_replay_equals_eager_bit_for_bit(hip_lib):
    """The recorded launch list (train_step.RecordedTrainStep: what engine.train_MTD_GAN_Ours replays from its third iteration
    on) against eager execution: five eager iterations == two eager + one recorded + two replayed, BIT-IDENTICAL -- parameters,
    spectral-norm vectors, optimizer moments, the 17 logged values of every iteration, with dropout ON (the masks are drawn
    from torch's generator at replay time) and a fresh batch per iteration (inputs are copied into the list's buffers).
    Between replays the test allocates and poisons device memory: a tensor the list still reads but no longer owns would pick
    that up.  Also through the API: train_MTD_GAN_Ours on a five-batch loader returns the same averages either way."""
    from mtd_gan_amd import engine, train_step as TS
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    dev = torch.device("cuda")
    batches = [tuple(t.cuda() for t in orc.synthetic_ldct(4, seed=70 + i)) for i in range(5)]

    def build():
        torch.manual_seed(5)
        m = MTD_GAN_Method().cuda().train()
        wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
        oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
        oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
        random.seed(123)
        torch.manual_seed(99)                                   # the dropout draws of the five iterations
        return m, wm, oD, oG

    def snapshot(m, oD, oG):
        st = {k: v.clone() for k, v in m.state_dict().items()}
        for tag, o in (("D", oD), ("G", oG)):
            for i, p in enumerate(o.param_groups[0]["params"]):
                if p in o.state:
                    st[f"{tag}.m{i}"], st[f"{tag}.v{i}"] = o.state[p]["exp_avg"].clone(), o.state[p]["exp_avg_sq"].clone()
        return st

    runs = {}
    for mode in ("eager", "list"):
        m, wm, oD, oG = build()
        logged = []
        for i, (x, y) in enumerate(batches):
            if mode == "eager":
                names, vals = engine.train_iteration(m, x, y, oG, oD, wm, None)
            else:
                names, vals = TS.recorded_iteration(m, x, y, oG, oD, wm, None)
                if i >= 2:
                    assert isinstance(m._mtd_recorded, TS.RecordedTrainStep)
                    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 10, 1 << 16, 1 << 20, 1 << 24, 3 << 24)]   # poison freed memory
                    del junk
            logged.append(vals.clone())
        torch.cuda.synchronize()
        runs[mode] = (snapshot(m, oD, oG), logged, oD.state[m.Discriminator.enc_out.weight]["step"], m)
    (sd_e, log_e, st_e, _), (sd_l, log_l, st_l, m_l) = runs["eager"], runs["list"]
    assert st_e == st_l == 5
    assert m_l._mtd_recorded.iterations == 3                  # recorded at the third iteration, two replays
    for i in range(5):
        assert torch.equal(log_l[i], log_e[i]), (i, log_l[i], log_e[i])
    for k in sd_e:
        assert torch.equal(sd_l[k], sd_e[k]), k
    # .grad of every parameter is what the last (replayed) iteration produced: usable by the caller as after an eager iteration
    m_e = runs["eager"][3]
    for (n, p), (_n, q) in zip(m_l.named_parameters(), m_e.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), n
    # ---- the API: same five batches through engine.train_MTD_GAN_Ours, list on / off
    stats = {}
    for mode in ("eager", "list"):
        m, wm, oD, oG = build()
        TS.LIST_MODE = mode == "list"
        try:
            stats[mode] = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y) for x, y in batches], oG, oD, dev, 0, 0, 4, wm)
            # an evaluation right after replays must see the UPDATED weights (no stale derived view outside the list)
            with torch.no_grad():
                stats[mode + "_pred"] = m.Generator(batches[0][0]).clone()
        finally:
            TS.LIST_MODE = True
    assert stats["eager"] == stats["list"], (stats["eager"], stats["list"])
    assert torch.equal(stats["eager_pred"], stats["list_pred"])
    # ---- an in-place update from OUTSIDE the loop (here: a checkpoint loaded into the live model) retires the list: the next
    # iterations are eager on the new weights, a new list is recorded at the third, and all of it equals plain eager execution
    ends = {}
    for mode in ("eager", "list"):
        m, wm, oD, oG = build()
        TS.LIST_MODE = mode == "list"
        try:
            for x, y in batches[:4]:
                TS.recorded_iteration(m, x, y, oG, oD, wm, None)
            first = getattr(m, "_mtd_recorded", None)
            with torch.no_grad():
                for p in m.Discriminator.parameters():
                    p.mul_(0.5)                         # (bumps the version counters, as load_state_dict's copy_ does)
            for x, y in batches:
                TS.recorded_iteration(m, x, y, oG, oD, wm, None)
            if mode == "list":
                assert isinstance(first, TS.RecordedTrainStep) and isinstance(m._mtd_recorded, TS.RecordedTrainStep) and m._mtd_recorded is not first
            torch.cuda.synchronize()
            ends[mode] = {k: v.clone() for k, v in m.state_dict().items()}
        finally:
            TS.LIST_MODE = True
    for k in ends["eager"]:
        assert torch.equal(ends["list"][k], ends["eager"][k]), k


def test_checkpoint_resume_like_train_py(hip_lib, tmp_path):
    """train.py:276-288 saves {model_state_dict, optimizer_D, scheduler_D, optimizer_G, scheduler_G, epoch}; train.py:146-159
    loads it on the CPU map and resumes.  A run that is saved and resumed that way continues exactly like the run that
    was never interrupted (same dropout masks and PCGrad orders injected)."""
    import random
    from mtd_gan_amd import engine
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    dev = torch.device("cuda")
    x, y = orc.synthetic_ldct(4, seed=77)
    batch = [dict(n_20=x, n_100=y)]

    def build(seed):
        torch.manual_seed(seed)
        m = MTD_GAN_Method().cuda().train()
        oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        sD = torch.optim.lr_scheduler.LambdaLR(oD, lr_lambda=lambda e: 0.9 ** e)
        sG = torch.optim.lr_scheduler.LambdaLR(oG, lr_lambda=lambda e: 0.9 ** e)
        return m, oD, oG, sD, sG, WeightMethods("pcgrad", n_tasks=3, device=dev)

    def epoch(m, oD, oG, sD, sG, wm, e):
        torch.manual_seed(1000 + e)           # dropout masks of the epoch
        random.seed(2000 + e)                 # PCGrad shuffle orders of the epoch
        engine.train_MTD_GAN_Ours(m, batch, oG, oD, dev, e, 0, 4, wm)
        sD.step()
        sG.step()

    a = build(1)
    epoch(*a, 0)
    ck = {"model_state_dict": a[0].state_dict(), "optimizer_D": a[1].state_dict(), "scheduler_D": a[3].state_dict(),
          "optimizer_G": a[2].state_dict(), "scheduler_G": a[4].state_dict(), "epoch": 0}
    path = str(tmp_path / "epoch_0_checkpoint.pth")
    torch.save(ck, path)
    epoch(*a, 1)                              # the uninterrupted run

    b = build(2)                              # different initial weights: everything must come from the file
    ck2 = torch.load(path, map_location="cpu")
    ck2["model_state_dict"] = {k.replace(".module", ""): v for k, v in ck2["model_state_dict"].items()}
    b[0].load_state_dict(ck2["model_state_dict"])
    b[1].load_state_dict(ck2["optimizer_D"])
    b[3].load_state_dict(ck2["scheduler_D"])
    b[2].load_state_dict(ck2["optimizer_G"])
    b[4].load_state_dict(ck2["scheduler_G"])
    assert ck2["epoch"] + 1 == 1
    epoch(*b, 1)
    sa, sb = a[0].state_dict(), b[0].state_dict()
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for oa, ob in ((a[1], b[1]), (a[2], b[2])):
        da, db = oa.state_dict(), ob.state_dict()
        assert da["param_groups"] == db["param_groups"]
        for i in da["state"]:
            for f in ("step", "exp_avg", "exp_avg_sq"):
                assert torch.equal(torch.as_tensor(da["state"][i][f]).cpu(), torch.as_tensor(db["state"][i][f]).cpu()), (i, f)


def test_torch_adamw_drives_the_same_step(hip_lib):
    """The reference's own optimizer (torch.optim.AdamW, train.py:122-126) instead of FusedAdamW: gradients are ordinary
    .grad tensors and the caches of derived weight views (packed conv weights, transposed mix weights, the generator tape)
    follow the tensors' version counters.  (a) one iteration lands where FusedAdamW lands (same gradients -- see the note at the
    comparison -- the update's fp32 rounding aside); (b) two iterations are bit-identical to two iterations with every cache dropped in between."""
    import random
    from mtd_gan_amd import engine, kernels as K
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    dev = torch.device("cuda")
    x, y = orc.synthetic_ldct(4, seed=5)

    def run(kind, iters, drop_caches=False):
        torch.manual_seed(3)
        m = MTD_GAN_Method().cuda().train()
        mk = (lambda ps: FusedAdamW(ps, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)) if kind == "fused" else \
             (lambda ps: torch.optim.AdamW(ps, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4))
        oD, oG = mk(m.Discriminator.parameters()), mk(m.Generator.parameters())
        wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
        for it in range(iters):
            torch.manual_seed(100 + it)
            random.seed(200 + it)
            engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, dev, it, 0, 4, wm)
            if drop_caches:
                K.weights_changed(None)
                m._gcache = None
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    f1, t1 = run("fused", 1), run("torch", 1)
    # A first AdamW step moves every weight by ~lr = 1e-4 whatever its gradient's size: lr * g / (|g| + eps).  The two routes take the
    # SAME gradients from the same kernels except the generator's 32 -> 32 channel 3x3 layers -- there the one-launch backward of the
    # deferred route multiplies directly where the plain-.grad route's weight gradient is the Winograd 32 x 32 kernel (plan 19), and the
    # two accumulate differently: 1e-6 of the tensor's largest element apart, which only an element whose gradient is down at AdamW's eps
    # (1e-8: these layers' gradients are ~1e-7 at this initialisation) can see.  So: (almost) every element within the update's own
    # fp32 rounding, at most three in a hundred beyond it, none by more than a whole step.  (Measured 0.3 % in round 5; 1.6 % on the last
    # block's conv in round 6, with the block convs on the Winograd kernels as well -- tools/r6_probe_routes.py: 126 of the 128 gradient
    # tensors differ between the routes by 1e-6 of their largest element under either set of kernels, each route by itself is repeatable
    # bit for bit, and these layers' gradients are 1e-15 ... 1e-9 at this initialisation, around and below AdamW's eps.)
    for k in f1:
        d = (f1[k].double() - t1[k].double()).abs()
        assert d.max().item() <= 2.1e-4, k
        assert (d > 2e-7).double().mean().item() <= (3e-2 if k.startswith("Generator.") and tuple(f1[k].shape) == (32, 32, 3, 3) else 0.0), (k, d.max().item())
    t2, t2_fresh = run("torch", 2), run("torch", 2, drop_caches=True)
    for k in t2:
        assert torch.equal(t2[k], t2_fresh[k]), k


def test_data_parallel_path_on_one_rank_equals_plain_step(hip_lib, monkeypatch):
    """The N > 1 code path (parallel.DataParallelSync: per-task all-reduce on a side stream before the PCGrad
    projection, task-specific and generator gradients bucketed) with every collective forced in a one-rank RCCL
    group (MTD_FORCE_DP=1): the average over one rank is the identity, so the iterations must give the plain
    step's parameters bit for bit -- the stream hand-offs around the collectives are what is being tested.
    Round 5: also THROUGH THE RECORDED LIST (two eager iterations, one recorded, five replayed -- collectives, the broadcast of the
    projection order and the hand-offs re-issued by the list on a real RCCL communicator), with the replica comparison of the
    first replays live, and dropout on."""
    import torch.distributed as dist
    from mtd_gan_amd import engine, parallel, train_step as TS
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    monkeypatch.setenv("MTD_FORCE_DP", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        batches = [tuple(t.cuda() for t in orc.synthetic_ldct(4, seed=78 + i)) for i in range(8)]
        results = {}
        for mode in ("plain", "dp", "dp_list"):
            torch.manual_seed(6)
            m = MTD_GAN_Method().cuda().train()
            wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
            oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
            oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
            dp = None
            if mode != "plain":
                dp = parallel.DataParallelSync(dev)
                assert dp.force and dp.world == 1
                dp.broadcast_module(m)
                wm.method.dp = dp
            random.seed(321)
            torch.manual_seed(77)                       # the dropout draws
            logged = []
            for x, y in batches:
                if mode == "dp_list":
                    names, vals = TS.recorded_iteration(m, x, y, oG, oD, wm, dp)
                else:
                    names, vals = engine.train_iteration(m, x, y, oG, oD, wm, dp)
                logged.append(vals.clone())
            torch.cuda.synchronize()
            if mode == "dp_list":
                st = m._mtd_recorded
                assert isinstance(st, TS.RecordedTrainStep) and st.iterations == 6, getattr(m, "_mtd_list_error", None)     # recorded + five replays
                assert st.replica_checks_left == 0 and m._mtd_list_error is None
                assert sum(1 for f, _a in st.list.ops if getattr(f, "mtd_collective", False) or getattr(getattr(f, "fn", None), "mtd_collective", False)) >= 5   # collectives are in the list
            results[mode] = ({k: v.clone() for k, v in m.state_dict().items()}, logged)
        sd_p, log_p = results["plain"]
        for mode in ("dp", "dp_list"):
            sd, log = results[mode]
            for i in range(len(batches)):
                assert torch.equal(log[i], log_p[i]), (mode, i)
            for k in sd_p:
                assert torch.equal(sd_p[k], sd[k]), (mode, k)
        # replicas_agree on this communicator: True on equal tensors (one rank), and the checksums see a single flipped bit
        from mtd_gan_amd import kernels as K
        t = torch.randn(1 << 20, device=dev)
        c0 = K.checksum_multi([t, t[:1000].contiguous()])
        t2 = t.clone()
        t2.view(torch.int32)[12345] ^= 1
        c1 = K.checksum_multi([t2, t2[:1000].contiguous()])
        want = t.view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF).sum()
        assert c0[0].item() == want.item() and c0[0].item() != c1[0].item() and c0[1].item() == c1[1].item()
        assert parallel.DataParallelSync(dev).replicas_agree([t, t2])
    finally:
        if created:
            dist.destroy_process_group()


def test_unusable_recording_falls_back_to_eager_without_losing_the_iteration(hip_lib, monkeypatch):
    """A recording that cannot be replayed (here: made to look as if the recorded iteration never read the PCGrad order slot)
    must not raise out of the training loop, must not run the batch twice, and must leave every later iteration eager: five
    iterations equal five plain eager ones bit for bit, with a warning that says why (`model._mtd_list_error`).  An optimizer
    state that is REPLACED while a list is live (optimizer.load_state_dict) retires the list instead of letting the replays
    write the old moment tensors (advisor, round 4)."""
    import warnings
    from mtd_gan_amd import engine, kernels as K, train_step as TS
    from mtd_gan_amd.arch.Ours.networks import MTD_GAN_Method
    from mtd_gan_amd.module.weight_methods import WeightMethods
    from mtd_gan_amd.optimizers import FusedAdamW
    dev = torch.device("cuda")
    batches = [tuple(t.cuda() for t in orc.synthetic_ldct(4, seed=170 + i)) for i in range(5)]

    def build():
        torch.manual_seed(15)
        m = MTD_GAN_Method().cuda().train()
        wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
        oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, weight_decay=5e-4)
        oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, weight_decay=5e-4)
        random.seed(9)
        torch.manual_seed(19)
        return m, wm, oD, oG

    m, wm, oD, oG = build()
    want = [engine.train_iteration(m, x, y, oG, oD, wm, None)[1].clone() for x, y in batches]
    sd_want = {k: v.clone() for k, v in m.state_dict().items()}

    m, wm, oD, oG = build()
    real_slot = TS.orders_slot

    def other_slot(d):          # the real slot while the iteration is being recorded, afterwards one the list has never seen
        return real_slot(d) if K.RECORDING is not None else K.HostScalars(d, 16, torch.int32)
    got = []
    for i, (x, y) in enumerate(batches):
        if i == 2:              # the recording iteration: RecordedTrainStep looks its slot up after the iteration has run
            monkeypatch.setattr(TS, "orders_slot", other_slot)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got.append(TS.recorded_iteration(m, x, y, oG, oD, wm, None)[1].clone())
        if i == 2:
            monkeypatch.setattr(TS, "orders_slot", real_slot)
            assert any("recorded launch list is not used" in str(x_.message) for x_ in w), [str(x_.message) for x_ in w]
            assert "order slot" in m._mtd_list_error and not isinstance(m._mtd_recorded, TS.RecordedTrainStep)
            assert oD.graph_mode is False and oG.graph_mode is False
    for i in range(5):
        assert torch.equal(got[i], want[i]), i
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd_want[k]), k
    assert oD.state[m.Discriminator.enc_out.weight]["step"] == 5        # every batch applied exactly once

    # ---- a replaced optimizer state retires a live list
    m, wm, oD, oG = build()
    for x, y in batches[:4]:
        TS.recorded_iteration(m, x, y, oG, oD, wm, None)
    first = m._mtd_recorded
    assert isinstance(first, TS.RecordedTrainStep)
    import copy
    oD.load_state_dict(copy.deepcopy(oD.state_dict()))                   # new moment tensors, same values (a checkpoint resume)
    assert not first.matches(m, oG, oD, wm, batches[0][0], batches[0][1], None)
    TS.recorded_iteration(m, *batches[4], oG, oD, wm, None)
    assert m._mtd_recorded is not first
    m2, wm2, oD2, oG2 = build()
    for x, y in batches:
        engine.train_iteration(m2, x, y, oG2, oD2, wm2, None)
    torch.cuda.synchronize()
    for (k, v), (_k, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(v, v2), k
    # the derived-view buffers of a model can be released explicitly (kernels.release_views)
    assert K.release_views(list(m2.parameters())) > 0

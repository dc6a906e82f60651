"""GPU parity of the ablation family (reference arch/Ours/networks.py:478-1937): RED-CNN generator, the five
partial discriminators, the ten Ablation_* wrappers -- losses, logged details and parameter gradients of the HIP path
against tests/golden/ablation.json (produced by the reference classes themselves) and the float64 oracle, and one
engine.train_MTD_GAN_Ours iteration with method_D=None (engine.py:56-73)."""
import json
import os

import pytest
import torch

import mtdgan_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


def _sample_idx(numel, k):
    return [(i * 2654435761 + 12345) % numel for i in range(k)]


def _check_elements(tag, named_grads, gold):
    """Sampled ELEMENTS of every gradient tensor against tests/golden/ablation_grad_samples.json (oracle/pin_grad_samples.py):
    the float64 value at each sampled position, bound max(1e-3, 2 x the reference's own fp32-vs-float64 error on that tensor)
    relative to the tensor's max-abs.  A gradient with the right norm but permuted or mis-scattered entries fails here."""
    bad = []
    for n, g in named_grads:
        e = gold[n]
        flat = g.detach().reshape(-1)
        idx = _sample_idx(flat.numel(), len(e["f64"]))
        got = flat[torch.tensor(idx, device=flat.device)].double().cpu().tolist()
        bound = max(TOL, 2 * e["err32"]) * e["maxabs"]
        for j, (a, b) in enumerate(zip(got, e["f64"])):
            if abs(a - b) > bound + 1e-30:
                bad.append((tag, n, j, a, b, bound))
        assert abs(flat.abs().max().item() - e["maxabs"]) <= 5e-3 * e["maxabs"] + 1e-30, (tag, n)
    assert not bad, bad[:8]


def _build(name, c):
    from mtd_gan_amd.arch.Ours import networks as N
    m = getattr(N, name)()
    gsd, dsd = m.Generator.state_dict(), m.Discriminator.state_dict()
    assert list(gsd.keys()) == c["g_keys"] and [list(v.shape) for v in gsd.values()] == c["g_shapes"], name      # the reference's keys
    assert list(dsd.keys()) == c["d_keys"] and [list(v.shape) for v in dsd.values()] == c["d_shapes"], name
    gst = orc.seeded_fill(dict(zip(c["g_keys"], map(tuple, c["g_shapes"]))), seed=c["gfill"])
    dst = orc.seeded_fill(dict(zip(c["d_keys"], map(tuple, c["d_shapes"]))), seed=c["dfill"])
    m.Generator.load_state_dict(gst)
    m.Discriminator.load_state_dict(dst)
    m.cuda().train()
    g = torch.Generator().manual_seed(c["mask_seed"])
    masks = [(torch.rand(2, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    return m, gst, dst, masks


@pytest.mark.parametrize("name", list(orc.ABLATIONS.keys()))
def test_ablation_wrapper_vs_reference_vectors(hip_lib, name):
    c = json.load(open(os.path.join(GOLD, "ablation.json")))[name]
    m, gst, dst, masks = _build(name, c)
    x, y = orc.synthetic_ldct(2, seed=c["data_seed"])
    xd, yd = x.cuda(), y.cuda()
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    d_tot, d_det = m.d_loss(xd, yd)
    assert list(d_det.keys()) == list(c["d_details"].keys()), name
    assert abs(d_tot.item() - c["d_loss"]) <= TOL * abs(c["d_loss"]), (name, d_tot.item(), c["d_loss"])
    for k, v in d_det.items():
        assert abs(v.item() - c["d_details"][k]) <= TOL * abs(c["d_details"][k]) + 2e-7, (name, k, v.item(), c["d_details"][k])
    d_tot.backward()
    torch.cuda.synchronize()
    bad = []
    for n, p in m.Discriminator.named_parameters():
        if n in c["d_no_grad"]:
            assert p.grad is None, (name, n)
            continue
        want = c["d_grad_norms"][n]
        if abs(p.grad.double().norm().item() - want) > 5e-3 * want + 1e-12:
            bad.append((n, p.grad.double().norm().item(), want))
    assert not bad, (name, bad[:6])
    gs = json.load(open(os.path.join(GOLD, "ablation_grad_samples.json")))[name]
    assert set(gs["d"].keys()) == {n for n, p in m.Discriminator.named_parameters() if p.grad is not None}, name
    _check_elements(name + " d_loss", [(n, p.grad) for n, p in m.Discriminator.named_parameters() if p.grad is not None], gs["d"])
    for p in m.parameters():
        p.grad = None
    g_tot, g_det = m.g_loss(xd, yd)
    assert list(g_det.keys()) == list(c["g_details"].keys()), name
    assert abs(g_tot.item() - c["g_loss"]) <= TOL * abs(c["g_loss"]), (name, g_tot.item(), c["g_loss"])
    for k, v in g_det.items():
        assert abs(v.item() - c["g_details"][k]) <= TOL * abs(c["g_details"][k]) + 2e-7, (name, k)
    g_tot.backward()
    torch.cuda.synchronize()
    bad = []
    for n, p in m.Generator.named_parameters():
        want = c["g_grad_norms"][n]
        if abs(p.grad.double().norm().item() - want) > 5e-3 * want + 1e-12:
            bad.append((n, p.grad.double().norm().item(), want))
    assert not bad, (name, bad[:6])
    _check_elements(name + " g_loss", list((n, p.grad) for n, p in m.Generator.named_parameters()), gs["g"])


def test_redcnn_generator_gradients_vs_fp64_oracle(hip_lib):
    """REDCNN_Generator forward and every parameter gradient against the float64 oracle (the criterion of the Res-FFT
    generator's test: error <= max(1e-3, 2 x the fp32 CPU path's own error against float64), per tensor)."""
    from mtd_gan_amd.arch.Ours.networks import REDCNN_Generator
    G = REDCNN_Generator(1, 32, 10, 3, 1)
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    st = orc.seeded_fill(shapes, seed=7)
    G.load_state_dict(st)
    G.cuda()
    x, y = orc.synthetic_ldct(3, seed=5)
    out = G(x.cuda())
    ((out - y.cuda()) ** 2).mean().backward()
    torch.cuda.synchronize()
    refs = {}
    for dt in (torch.float64, torch.float32):
        s = {k: v.to(dt).clone().requires_grad_(True) for k, v in st.items()}
        o = orc.redcnn_forward(s, x.to(dt))
        ((o - y.to(dt)) ** 2).mean().backward()
        refs[dt] = (o.detach(), s)
    o64, s64 = refs[torch.float64]
    assert (out.detach().cpu().double() - o64).abs().max().item() <= TOL * o64.abs().max().item()
    for n, p in G.named_parameters():
        r = s64[n].grad
        den = r.abs().max().item() + 1e-30
        e_hip = (p.grad.cpu().double() - r).abs().max().item() / den
        e_cpu = (refs[torch.float32][1][n].grad.double() - r).abs().max().item() / den
        assert e_hip <= max(TOL, 2 * e_cpu), (n, e_hip, e_cpu)


def test_ablation_training_iteration_without_weight_method(hip_lib):
    """engine.train_MTD_GAN_Ours with method_D=None (engine.py:56-73: d_loss.backward() on the wrapper's scalar): one
    iteration of Ablation_CLS_SEG_REC_NDS_RC against the oracle's own plain-backward step -- logged scalars and updated
    parameters."""
    from mtd_gan_amd import engine
    from mtd_gan_amd.optimizers import FusedAdamW
    name = "Ablation_CLS_SEG_REC_NDS_RC"
    c = json.load(open(os.path.join(GOLD, "ablation.json")))[name]
    m, gst, dst, masks = _build(name, c)
    m.Discriminator._inject_masks = [k.clone() for k in masks]
    x, y = orc.synthetic_ldct(2, seed=c["data_seed"])
    oD = FusedAdamW(m.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oG = FusedAdamW(m.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    stats = engine.train_MTD_GAN_Ours(m, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cuda"), 0, 0, 2, None)
    assert abs(stats["d_loss"] - c["d_loss"]) <= TOL * abs(c["d_loss"])
    for k, v in c["d_details"].items():
        assert abs(stats[k] - v) <= TOL * abs(v) + 2e-6, (k, stats[k], v)
    # the G step sees the updated discriminator: compare with the oracle stepping the same way
    st = {"Generator." + k: v.clone() for k, v in gst.items()}
    st.update({"Discriminator." + k: v.clone() for k, v in dst.items()})
    dn = ["Discriminator." + n for n, _ in m.Discriminator.named_parameters()]
    for n in dn:
        st[n] = st[n].requires_grad_(True)
    tot, _ = orc.ablation_losses(name, st, x, y, masks[:4], "d")
    grads = torch.autograd.grad(tot, [st[n] for n in dn])
    for n, g in zip(dn, grads):
        p, _, _ = orc.adamw_step(st[n].detach(), g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-4)
        st[n] = p
    st = {k: v.detach() for k, v in st.items()}
    with torch.no_grad():
        g_tot, g_det = orc.ablation_losses(name, st, x, y, masks[4:5], "g")
    assert abs(stats["g_loss"] - float(g_tot)) <= TOL * abs(float(g_tot)), (stats["g_loss"], float(g_tot))
    post = m.state_dict()
    for k in ("Discriminator.conv11.weight_u", "Discriminator.r_dconv62.weight_v", "Discriminator.c_fc.weight_u"):
        assert (post[k].cpu() - st[k]).abs().max().item() <= TOL * st[k].abs().max().item(), k

"""CPU tests: the oracle (oracle/mtdgan_oracle.py) against the committed golden vectors that
oracle/pin_against_reference.py generated from the real reference.  Runs anywhere (no GPU, no reference)."""
import json
import os
import random

import numpy as np
import torch

import mtdgan_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def test_block_matches_reference_vectors():
    z = np.load(os.path.join(GOLD, "block.npz"))
    st = {k: v.clone().requires_grad_(True) for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items() if k.startswith("enforce.0.")}
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(2, 32, 64, 64, generator=gen) * 0.5).requires_grad_(True)
    out = orc._blk(st, 0, x)
    cot = torch.randn(out.shape, generator=gen)
    (out * cot).sum().backward()
    assert rel(out.detach()[:, ::8, ::4, ::4], z["out_sample"]) < 1e-5
    assert abs(out.detach().double().sum().item() - float(z["out_sum"])) < 1e-3 * float(z["out_abs"]) * 1e-3
    assert rel(x.grad[:, ::8, ::4, ::4], z["dx_sample"]) < 1e-5
    assert rel(st["enforce.0.fft_conv.weight"].grad, z["g_fft_conv_weight"]) < 1e-4
    assert rel(st["enforce.0.img_conv.weight"].grad.double().norm(), z["gn_img_conv_weight"]) < 1e-5
    # the explicit irfft2 formula (what the HIP rows kernel implements) agrees as well
    out_e = orc._blk(st, 0, x, explicit_irfft=True)
    assert rel(out_e.detach(), out.detach()) < 1e-5


def test_generator_matches_reference_vectors():
    z = np.load(os.path.join(GOLD, "generator.npz"))
    gs = {k: v.clone().requires_grad_(True) for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=7).items()}
    x, y = orc.synthetic_ldct(2, seed=1234)
    assert rel(x[0, 0, 0, :4], z["x_head"]) < 1e-6
    out = orc.generator_forward(gs, x)
    assert rel(out.detach(), z["out"]) < 1e-5
    (out * torch.from_numpy(z["cot"])).sum().backward()
    for n, gn in zip(z["grad_names"], z["grad_norms"]):
        assert abs(gs[str(n)].grad.double().norm().item() - gn) <= 1e-4 * gn + 1e-12, n
    assert abs(float(orc.psnr(out.detach().clip(0, 1), y)) - float(z["psnr"])) < 1e-3
    assert abs(float(orc.ssim(out.detach().clip(0, 1), y)) - float(z["ssim"])) < 1e-5
    assert abs(float(orc.rmse(out.detach().clip(0, 1), y)) - float(z["rmse"])) < 1e-6


def test_discriminator_matches_reference_vectors():
    z = np.load(os.path.join(GOLD, "discriminator.npz"))
    dstate = orc.seeded_fill(orc.d_state_shapes(), seed=9)
    _, y = orc.synthetic_ldct(2, seed=1234)
    with torch.no_grad():
        e, s, r = orc.discriminator_forward(dict(dstate), y, train=False)
    assert rel(e, z["eval_enc"]) < 1e-5 and rel(s, z["eval_dec"]) < 1e-5 and rel(r, z["eval_rec"]) < 1e-5
    g = torch.Generator().manual_seed(21)
    masks = [(torch.rand(2, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(2)]
    ds = {k: v.clone() for k, v in dstate.items()}
    names = orc.d_shared_names() + orc.d_task_specific_names() + ["c_fc.bias", "c_fc.weight_orig"]
    for n in names:
        ds[n] = ds[n].requires_grad_(True)
    leaves = {n: ds[n] for n in names}
    yin = y.clone().requires_grad_(True)
    for it in range(2):
        outs = orc.discriminator_forward(ds, yin, train=True, drop_mask=masks[it])
    assert rel(outs[0].detach(), z["train_enc"]) < 1e-5
    assert rel(outs[1].detach(), z["train_dec"]) < 1e-5
    assert rel(outs[2].detach(), z["train_rec"]) < 1e-5
    cots = [torch.from_numpy(z[k]) for k in ("cot_enc", "cot_dec", "cot_rec")]
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    assert rel(yin.grad, z["dinput"]) < 1e-4
    for n, gn in zip(z["grad_names"], z["grad_norms"]):
        assert abs(leaves[str(n)].grad.double().norm().item() - gn) <= 1e-4 * gn + 1e-12, n
    assert rel(ds["conv11.weight_u"], z["u_conv11"]) < 1e-5
    assert rel(ds["bconv2.weight_u"], z["u_bconv2"]) < 1e-5
    assert rel(ds["c_fc.weight_v"], z["v_c_fc"]) < 1e-5


def test_pcgrad_gram_form_matches_reference_vectors():
    cases = json.load(open(os.path.join(GOLD, "pcgrad.json")))
    for c in cases:
        random.seed(c["seed"])
        assert orc.shuffle_orders(3) == c["orders"]
        w = orc.pcgrad_coefficients(c["gram"], c["orders"])
        assert max(abs(a - b) for a, b in zip(w, c["w"])) < 1e-12


def test_full_step_matches_reference_vectors():
    z = json.load(open(os.path.join(GOLD, "step_seeded.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"]).items()})
    x, y = orc.synthetic_ldct(z["batch"], seed=z["data_seed"])
    g = torch.Generator().manual_seed(z["mask_seed"])
    masks = [(torch.rand(z["batch"], 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    res = orc.train_step(full, {}, x, y, masks, z["orders"], lr=z["lr"])
    assert rel(res["d_losses"], z["d_losses"]) < 1e-5
    assert abs(res["g_loss"] - z["stats"]["g_loss"]) < 1e-4
    for k, v in {**res["d_details"], **res["g_details"]}.items():
        assert abs(v - z["stats"][k]) <= 2e-4 * abs(z["stats"][k]) + 2e-7, k
    assert rel(torch.tensor(res["gram"]), torch.tensor(z["gram"])) < 1e-6
    for k, samples in z["post_samples"].items():
        t = full[k].reshape(-1)
        n = t.numel()
        for i, s in enumerate(samples):
            idx = (i * 2654435761 + 12345) % n
            assert abs(t[idx].item() - s) <= 1e-5 * abs(s) + 1e-8, (k, i)
    assert torch.equal(full["Discriminator.c_fc.weight_orig"],
                       orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"])["c_fc.weight_orig"])   # frozen (quirk 1)


def test_whole_slice_generator_and_pixel_metrics_golden():
    """Whole-slice inference (engine.py:89,129) and metrics.py:172-244 of the oracle against the vectors generated from the reference."""
    z = np.load(os.path.join(GOLD, "generator128.npz"))
    g = orc.seeded_fill(orc.g_param_shapes(), seed=7)
    x, y = orc.synthetic_ldct(1, seed=77, size=128)
    assert np.allclose(x[0, 0, 0, :4].numpy(), z["x_head"])
    with torch.no_grad():
        out = orc.generator_forward(g, x)
    assert np.abs(out.numpy() - z["out"]).max() < 2e-6
    pred = out.clip(0, 1)
    for i, a in enumerate((x, y, pred)):
        assert abs(orc.psnr(a, y).item() - z["psnr"][i]) < 1e-4
        assert abs(orc.ssim(a, y).item() - z["ssim"][i]) < 1e-5
        assert abs(orc.rmse(a, y).item() - z["rmse"][i]) < 1e-6


def test_pcgrad_optimizer_wrapper_matches_reference_vectors():
    """module/pcgrad.py::PCGrad (a13b): the oracle's restatement on the reference's own self-test nets."""
    import torch.nn as nn
    cases = json.load(open(os.path.join(GOLD, "pcgrad_wrapper.json")))
    assert [c["tag"] for c in cases] == ["TestNet", "MultiHeadTestNet", "MultiHeadTestNet/3 objectives"]
    for c in cases:
        st = {k: torch.tensor(v, requires_grad=True) for k, v in c["state"].items()}
        x, ys = torch.tensor(c["x"]), [torch.tensor(y) for y in c["ys"]]
        lin = lambda pre, t: torch.nn.functional.linear(t, st[pre + ".weight"], st[pre + ".bias"])
        if c["tag"] == "TestNet":
            objs = [nn.L1Loss()(lin("_linear", x), ys[0]), nn.MSELoss()(lin("_linear", x), ys[0])]
        else:
            feat = lin("_linear", x)
            h1, h2 = lin("_head1", feat), lin("_head2", feat)
            objs = [nn.MSELoss()(h1, ys[0]), nn.MSELoss()(h2, ys[0])] if len(ys) == 1 else \
                   [nn.MSELoss()(h1, ys[0]), nn.MSELoss()(h1, -ys[0] + 0.1), nn.L1Loss()(h2, ys[1])]
        params = list(st.values())
        flat, has = [], []
        for obj in objs:
            gs = torch.autograd.grad(obj, params, retain_graph=True, allow_unused=True)
            flat.append(torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]))
            has.append(torch.cat([(torch.ones_like(p) if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]))
        random.seed(c["rseed"])
        merged = orc.pcgrad_wrapper_merge(flat, has)
        ref = torch.cat([torch.tensor(c["grads"][k]).reshape(-1) for k in c["state"]])
        assert rel(merged, ref) < 1e-6, c["tag"]


def test_config1_replay_and_oracle_step():
    """BASELINE configs[0]: the reference's default-init state rebuilt from the recorded draw tape (checked against the
    reference's per-tensor checksums inside load()), then the oracle's step on it against the reference's 17 scalars."""
    import replay_config1
    z, state, x, y, masks = replay_config1.load()
    random.seed(z["seed"])
    orders = orc.shuffle_orders(3)
    assert orders == z["orders"]
    res = orc.train_step(state, {}, x, y, masks, orders, lr=z["lr"])
    got = {"d_loss": float(sum(res["d_losses"])), "g_loss": res["g_loss"], **res["d_details"], **res["g_details"]}
    for k, v in z["stats"].items():
        if k != "lr":
            assert abs(got[k] - v) <= 1e-5 * abs(v) + 2e-7, (k, got[k], v)


def test_full_step_b32_matches_reference_vectors():
    """BASELINE configs[2] size: the seeded-fill step at 32 patches and the generator's PSNR after the step."""
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"]).items()})
    x, y = orc.synthetic_ldct(z["batch"], seed=z["data_seed"])
    g = torch.Generator().manual_seed(z["mask_seed"])
    masks = [(torch.rand(z["batch"], 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
    res = orc.train_step(full, {}, x, y, masks, z["orders"], lr=z["lr"])
    assert rel(res["d_losses"], z["d_losses"]) < 1e-5
    for k, v in {**res["d_details"], **res["g_details"]}.items():
        assert abs(v - z["stats"][k]) <= 2e-4 * abs(z["stats"][k]) + 2e-7, k
    with torch.no_grad():
        pred = orc.generator_forward(full, x, "Generator.")
    assert abs(float(orc.psnr(pred.clip(0, 1), y)) - z["post_metrics"]["psnr"]) < 1e-4
    assert rel(pred[:, 0, ::16, ::16], torch.tensor(z["post_pred_sample"])) < 1e-5
    # the G step's gradients: the oracle's per-parameter norms against the REFERENCE's own .grad norms (round 5:
    # g_grad_samples_b32.json holds what engine.py:50-52 left in the generator's .grad)
    gs = json.load(open(os.path.join(GOLD, "g_grad_samples_b32.json")))
    assert len(gs["grads"]) == 128 and set(gs["grads"]) == set(res["g_grad_norms"])
    for n, e in gs["grads"].items():
        assert abs(res["g_grad_norms"][n] - e["norm_ref32"]) <= 1e-4 * e["norm_ref32"] + 1e-12, n
        assert abs(z["g_grad_norms"][n] - e["norm_ref32"]) <= 1e-4 * e["norm_ref32"] + 1e-12, n


def test_ablation_family_matches_reference_vectors():
    """The ten ablation wrappers (networks.py:1324-1937) of the oracle against the losses the reference classes produced."""
    z = json.load(open(os.path.join(GOLD, "ablation.json")))
    assert list(z.keys()) == list(orc.ABLATIONS.keys())
    x, y = orc.synthetic_ldct(2, seed=1234)
    for name in ("Ablation_SEG", "Ablation_CLS_REC", "Ablation_CLS_SEG_REC_NDS_RC_ResFFT"):      # (the whole family: the GPU test)
        c = z[name]
        st = {"Generator." + k: v for k, v in orc.seeded_fill(dict(zip(c["g_keys"], map(tuple, c["g_shapes"]))), seed=c["gfill"]).items()}
        st.update({"Discriminator." + k: v for k, v in orc.seeded_fill(dict(zip(c["d_keys"], map(tuple, c["d_shapes"]))), seed=c["dfill"]).items()})
        g = torch.Generator().manual_seed(c["mask_seed"])
        masks = [(torch.rand(2, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5)]
        npass = 4 if orc.ABLATIONS[name][5] else 2
        with torch.no_grad():
            d_tot, d_det = orc.ablation_losses(name, st, x, y, masks[:npass], "d")
            g_tot, g_det = orc.ablation_losses(name, st, x, y, masks[npass:npass + 1], "g")
        assert abs(float(d_tot) - c["d_loss"]) <= 1e-5 * abs(c["d_loss"]) and abs(float(g_tot) - c["g_loss"]) <= 1e-5 * abs(c["g_loss"]), name
        assert list(d_det.keys()) == list(c["d_details"].keys()) and list(g_det.keys()) == list(c["g_details"].keys()), name
        for k, v in {**d_det, **g_det}.items():
            want = {**c["d_details"], **c["g_details"]}[k]
            assert abs(float(v) - want) <= 2e-5 * abs(want) + 1e-9, (name, k)


def test_three_iterations_match_reference_vectors():
    """tests/golden/step_seeded_x3.json (oracle/pin_three_iterations.py): the reference's engine.train_MTD_GAN_Ours on three
    batches in a row -- AdamW at steps 1-3, u / v after 15 discriminator forwards.  The oracle restatement reproduces the 17
    logged scalars of EVERY iteration and the sampled state after every iteration (an Adam update is ~ +-lr whatever the
    gradient's size: updates within 0.05 lr, at most 1 % of the samples may miss -- elements whose gradient is rounding noise)."""
    z = json.load(open(os.path.join(GOLD, "step_seeded_x3.json")))
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z["dfill"]).items()})
    st = {k: v.clone() for k, v in full.items()}
    g = torch.Generator().manual_seed(z["mask_seed"])
    masks = [(torch.rand(z["batch"], 512, generator=g) >= 0.3).float() / 0.7 for _ in range(5 * z["iters"])]
    random.seed(z["random_seed"])
    opt = {}
    for i in range(z["iters"]):
        x, y = orc.synthetic_ldct(z["batch"], seed=z["data_seed"] + i)
        orders = orc.shuffle_orders(3)
        assert orders == z["orders"][i]
        res = orc.train_step(st, opt, x, y, masks[5 * i:5 * i + 5], orders, lr=z["lr"])
        got = {"d_loss": float(sum(res["d_losses"])), "g_loss": float(res["g_loss"])}
        got.update({k: float(v) for k, v in res["d_details"].items()})
        got.update({k: float(v) for k, v in res["g_details"].items()})
        for k, v in got.items():
            want = z["per_iter"][i][k]
            assert abs(v - want) <= 1e-4 * abs(want) + 2e-7, (i, k, v, want)
        bad, n = 0, 0
        for k, samples in z["post_samples"][i].items():
            t = st[k].detach().reshape(-1)
            for j, sv in enumerate(samples):
                idx = (j * 2654435761 + 12345) % t.numel()
                tol = 1e-3 * max(abs(sv), 1e-3) if k.endswith(("weight_u", "weight_v")) else 0.05 * z["lr"] + 1e-9
                bad += abs(t[idx].item() - sv) > tol
                n += 1
        assert bad <= 0.01 * n, (i, bad, n)
    for k, v in z["loader_stats"].items():          # the three-batch loader's averages are the mean of the three iterations
        if k != "lr":
            mean = sum(s[k] for s in z["per_iter"]) / z["iters"]
            assert abs(v - mean) <= 1e-6 * abs(mean) + 2e-7

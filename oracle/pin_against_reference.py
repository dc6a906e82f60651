"""Pin oracle/mtdgan_oracle.py to the real reference and write tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference).  Usage:  python oracle/pin_against_reference.py
Every check asserts oracle == reference (fp32 CPU, tolerance printed) before a vector is written.
The vectors are data: seeded inputs (by recipe), expected outputs / norms / sampled elements.
"""
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import _refboot  # noqa: E402
import rng_tape  # noqa: E402

_refboot.boot()
import mtdgan_oracle as orc  # noqa: E402
from arch.Ours.networks import (FFT_ConvBlock, MTD_GAN_Method, Multi_Task_Discriminator_Skip,  # noqa: E402
                                ResFFT_Generator)
from module.weight_methods import WeightMethods  # noqa: E402
import engine  # noqa: E402
import metrics  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def close(a, b, tol, what, atol=0.0):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    rel = err / ref
    print(f"  {what:<44s} max|d|={err:.3e}  rel={rel:.3e}")
    assert rel <= tol or err <= atol, (what, err, rel)
    return rel


class RecDrop(torch.nn.Module):
    """Stands in for nn.Dropout(p): draws the mask with the same RNG call F.dropout would make on a
    tensor of this shape, records it, or replays injected masks.  Result == the reference's dropout."""

    def __init__(self, p=0.3, inject=None):
        super().__init__()
        self.p = p
        self.inject = list(inject) if inject is not None else None
        self.rec = []

    def forward(self, x):
        if not self.training:
            return x
        if self.inject is not None:
            m = self.inject.pop(0)
        else:
            m = F.dropout(torch.ones_like(x), self.p, True)
        self.rec.append(m.clone())
        return x * m


def mask_seq(n, batch, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(batch, 512, generator=g) >= 0.3).float() / 0.7 for _ in range(n)]


def sample_idx(numel, k=8):
    return [(i * 2654435761 + 12345) % numel for i in range(k)]


def main():
    t0 = time.time()
    # ------------------------------------------------------------------ 0. key / shape contract
    ref = MTD_GAN_Method()
    rsd = ref.state_dict()
    gshapes, dshapes = orc.g_param_shapes(), orc.d_state_shapes()
    mine = {"Generator." + k: tuple(v) for k, v in gshapes.items()}
    mine.update({"Discriminator." + k: tuple(v) for k, v in dshapes.items()})
    assert set(mine) == set(rsd.keys()), set(mine) ^ set(rsd.keys())
    for k, v in rsd.items():
        assert tuple(v.shape) == mine[k], (k, v.shape, mine[k])
    assert [n for n, _ in ref.Discriminator.named_parameters() if any(p is q for q in ref.Discriminator.shared_parameters() for p in [dict(ref.Discriminator.named_parameters())[n]])] == orc.d_shared_names()
    npar = dict(ref.Discriminator.named_parameters())
    ts_ref = [next(n for n, q in npar.items() if q is p) for p in ref.Discriminator.task_specific_parameters()]
    assert ts_ref == orc.d_task_specific_names(), "task-specific order"
    print("state_dict contract: 326 keys / shapes / partition order OK")

    # ------------------------------------------------------------------ 1. Res-FFT-Conv block
    print("block")
    gstate = orc.seeded_fill(gshapes, seed=7)
    blk = FFT_ConvBlock(32)
    blk.load_state_dict({k[len("enforce.0."):]: v for k, v in gstate.items() if k.startswith("enforce.0.")})
    gen = torch.Generator().manual_seed(11)
    xb = torch.randn(2, 32, 64, 64, generator=gen) * 0.5
    xb_r = xb.clone().requires_grad_(True)
    out_r = blk(xb_r)
    cot = torch.randn(out_r.shape, generator=gen)
    (out_r * cot).sum().backward()
    st = {k: v.clone().requires_grad_(True) for k, v in gstate.items() if k.startswith("enforce.0.")}
    xb_o = xb.clone().requires_grad_(True)
    out_o = orc._blk(st, 0, xb_o)
    out_e = orc._blk(st, 0, xb_o, explicit_irfft=True)
    close(out_o, out_r, 1e-6, "block fwd")
    close(out_e, out_r, 2e-6, "block fwd (explicit irfft2 formula)")
    (out_o * cot).sum().backward()
    close(xb_o.grad, xb_r.grad, 1e-5, "block dx")
    bg = {}
    for n, p in blk.named_parameters():
        close(st["enforce.0." + n].grad, p.grad, 1e-5, "block d" + n)
        bg[n] = p.grad
    np.savez_compressed(os.path.join(GOLD, "block.npz"),
                        out_sample=out_r.detach()[:, ::8, ::4, ::4].numpy(), out_sum=out_r.detach().double().sum().item(),
                        out_abs=out_r.detach().double().abs().sum().item(),
                        dx_sample=xb_r.grad[:, ::8, ::4, ::4].numpy(), dx_norm=xb_r.grad.double().norm().item(),
                        **{"g_" + n.replace(".", "_"): g.numpy() for n, g in bg.items() if g.numel() <= 4096},
                        **{"gn_" + n.replace(".", "_"): g.double().norm().item() for n, g in bg.items()})

    # ------------------------------------------------------------------ 2. Generator fwd + bwd
    print("generator")
    G = ResFFT_Generator(1, 32, 10, 3, 1)
    G.load_state_dict(gstate)
    x2, y2 = orc.synthetic_ldct(2, seed=1234)
    out_r = G(x2)
    gs = {k: v.clone().requires_grad_(True) for k, v in gstate.items()}
    out_o = orc.generator_forward(gs, x2)
    close(out_o, out_r, 2e-6, "G fwd")
    cot = torch.randn(out_r.shape, generator=gen)
    (out_r * cot).sum().backward()
    (out_o * cot).sum().backward()
    gn = {}
    worst = 0.0
    for n, p in G.named_parameters():
        d = (gs[n].grad - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-30)
        worst = max(worst, d)
        gn[n] = p.grad.double().norm().item()
    print(f"  G param grads worst rel {worst:.3e}")
    assert worst < 1e-4
    np.savez_compressed(os.path.join(GOLD, "generator.npz"), x_head=x2[0, 0, 0, :4].numpy(), out=out_r.detach().numpy(),
                        cot=cot.numpy(), grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values())),
                        psnr=float(metrics.compute_PSNR(x2, y2, out_r.detach().clip(0, 1))[2]),
                        ssim=float(metrics.compute_SSIM(x2, y2, out_r.detach().clip(0, 1))[2]),
                        rmse=float(metrics.compute_RMSE(x2, y2, out_r.detach().clip(0, 1))[2]))
    close(orc.psnr(out_o.detach().clip(0, 1), y2), metrics.compute_PSNR(x2, y2, out_r.detach().clip(0, 1))[2], 1e-6, "PSNR")
    close(orc.ssim(out_o.detach().clip(0, 1), y2), metrics.compute_SSIM(x2, y2, out_r.detach().clip(0, 1))[2], 1e-5, "SSIM")
    close(orc.rmse(out_o.detach().clip(0, 1), y2), metrics.compute_RMSE(x2, y2, out_r.detach().clip(0, 1))[2], 1e-6, "RMSE")
    x32, _ = orc.synthetic_ldct(32, seed=1234)
    close(x32[0, 0, 0, :4], torch.tensor([0.243470, 0.229047, 0.101727, 0.305131]), 1e-5, "synthetic_ldct anchor (SURVEY 8d)")

    # ------------------------------------------------------------------ 2b. whole-slice inference (engine.py:89,129), 128 x 128
    print("generator, whole-slice inference (1 x 1 x 128 x 128, no_grad)")
    x3, y3 = orc.synthetic_ldct(1, seed=77, size=128)
    with torch.no_grad():
        out_r3 = G(x3)
        out_o3 = orc.generator_forward(gstate, x3)
    close(out_o3, out_r3, 2e-6, "G fwd 128")
    m_r = [metrics.compute_PSNR(x3, y3, out_r3.clip(0, 1)), metrics.compute_SSIM(x3, y3, out_r3.clip(0, 1)), metrics.compute_RMSE(x3, y3, out_r3.clip(0, 1))]
    close(orc.psnr(x3, y3), m_r[0][0], 1e-6, "PSNR input")
    close(orc.ssim(x3, y3), m_r[1][0], 1e-5, "SSIM input")
    close(orc.ssim(y3, y3), m_r[1][1], 1e-6, "SSIM gt")
    close(orc.ssim(out_o3.clip(0, 1), y3), m_r[1][2], 1e-5, "SSIM pred")
    np.savez_compressed(os.path.join(GOLD, "generator128.npz"), x_head=x3[0, 0, 0, :4].numpy(), out=out_r3.numpy(),
                        psnr=np.array(m_r[0], dtype=np.float64), ssim=np.array(m_r[1], dtype=np.float64), rmse=np.array(m_r[2], dtype=np.float64))

    # ------------------------------------------------------------------ 3. Discriminator fwd (eval, train) + bwd
    print("discriminator")
    dstate = orc.seeded_fill(dshapes, seed=9)
    D = Multi_Task_Discriminator_Skip(1, 64)
    D.load_state_dict(dstate)
    D.eval()
    with torch.no_grad():
        e_r, s_r, r_r = D(y2)
        e_o, s_o, r_o = orc.discriminator_forward(dict(dstate), y2, train=False)
    close(e_o, e_r, 1e-5, "D eval enc")
    close(s_o, s_r, 1e-5, "D eval dec")
    close(r_o, r_r, 1e-5, "D eval rec")
    gold_d = dict(eval_enc=e_r.numpy(), eval_dec=s_r.numpy(), eval_rec=r_r.numpy())
    D.train()
    masks = mask_seq(2, 2, seed=21)
    D.c_drop = RecDrop(0.3, inject=masks)
    ds = {k: v.clone() for k, v in dstate.items()}
    for n in orc.d_shared_names() + orc.d_task_specific_names() + ["c_fc.bias", "c_fc.weight_orig"]:
        ds[n] = ds[n].requires_grad_(True)
    leaves = {n: ds[n] for n in orc.d_shared_names() + orc.d_task_specific_names() + ["c_fc.bias", "c_fc.weight_orig"]}
    yin_r = y2.clone().requires_grad_(True)
    yin_o = y2.clone().requires_grad_(True)
    cots = [torch.randn(2, 1, generator=gen), torch.randn(2, 1, 64, 64, generator=gen) * 0.1, torch.randn(2, 1, 64, 64, generator=gen) * 0.1]
    for it in range(2):   # two train forwards: u/v evolve
        outs_r = D(yin_r)
        outs_o = orc.discriminator_forward(ds, yin_o, train=True, drop_mask=masks[it])
        for a, b, nm in zip(outs_o, outs_r, ("enc", "dec", "rec")):
            close(a, b, 1e-5, f"D train#{it} {nm}")
    sum((o * c).sum() for o, c in zip(outs_r, cots)).backward()
    sum((o * c).sum() for o, c in zip(outs_o, cots)).backward()
    close(yin_o.grad, yin_r.grad, 1e-4, "D d/dinput")
    worst, dn = 0.0, {}
    for n, p in D.named_parameters():
        d = (leaves[n].grad - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-30)
        worst = max(worst, d)
        dn[n] = p.grad.double().norm().item()
    print(f"  D param grads worst rel {worst:.3e}")
    assert worst < 1e-4
    for n in ("conv11", "down3", "bconv2", "c_fc", "s_dconv31", "r_dconv62"):
        close(ds[n + ".weight_u"], D.state_dict()[n + ".weight_u"], 1e-5, f"u after 2 fwds {n}")
    gold_d.update(train_enc=outs_r[0].detach().numpy(), train_dec=outs_r[1].detach().numpy(), train_rec=outs_r[2].detach().numpy(),
                  cot_enc=cots[0].numpy(), cot_dec=cots[1].numpy(), cot_rec=cots[2].numpy(),
                  dinput=yin_r.grad.numpy(), grad_names=np.array(list(dn.keys())), grad_norms=np.array(list(dn.values())),
                  u_conv11=D.state_dict()["conv11.weight_u"].numpy(), u_bconv2=D.state_dict()["bconv2.weight_u"].numpy(),
                  v_c_fc=D.state_dict()["c_fc.weight_v"].numpy())
    np.savez_compressed(os.path.join(GOLD, "discriminator.npz"), **gold_d)

    # ------------------------------------------------------------------ 4. PCGrad projection vs Gram form
    print("pcgrad")
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cpu"))
    rng_cases = []
    for trial in range(50):
        g = torch.Generator().manual_seed(1000 + trial)
        base = torch.randn(3, 40, generator=g, dtype=torch.float64)
        if trial % 3 == 0:
            base[1] = -0.7 * base[0] + 0.3 * base[1]
        if trial % 5 == 0:
            base[2] = -base[1] * 0.2 + base[2] * 0.05
        grads = [(base[i, :25].clone(), base[i, 25:].clone()) for i in range(3)]
        random.seed(trial)
        state0 = random.getstate()
        merged_ref = wm.method._project_conflicting([tuple(t.clone() for t in g_) for g_ in grads])
        merged_ref = torch.cat([m.reshape(-1) for m in merged_ref])
        random.setstate(state0)
        orders = orc.shuffle_orders(3)
        gram = (base @ base.t()).tolist()
        w = orc.pcgrad_coefficients(gram, orders)
        merged_o = sum(wk * base[k] for k, wk in enumerate(w))
        random.setstate(state0)
        merged_l = orc.pcgrad_merge([base[i].clone() for i in range(3)])
        assert (merged_o - merged_ref).abs().max() < 1e-10 and (merged_l - merged_ref).abs().max() < 1e-10
        if trial < 8:
            rng_cases.append(dict(seed=trial, orders=orders, gram=gram, w=w))
    print("  50 trials: gram-form == literal == reference (1e-10)")
    with open(os.path.join(GOLD, "pcgrad.json"), "w") as f:
        json.dump(rng_cases, f)

    # ------------------------------------------------------------------ 5. Full step with seeded fill (B=2)
    print("full step (seeded fill, B=2, lr 1e-4)")
    full = {"Generator." + k: v for k, v in gstate.items()}
    full.update({"Discriminator." + k: v for k, v in dstate.items()})
    model = MTD_GAN_Method()
    model.load_state_dict(full)
    masks5 = mask_seq(5, 2, seed=33)
    model.Discriminator.c_drop = RecDrop(0.3, inject=masks5)
    oD = torch.optim.AdamW([dict(params=model.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4),
                            dict(params=wm.parameters(), lr=0.025, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)])
    oG = torch.optim.AdamW(model.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    random.seed(77)
    rstate = random.getstate()
    stats = engine.train_MTD_GAN_Ours(model, [dict(n_20=x2, n_100=y2)], oG, oD, torch.device("cpu"), 0, 1, 2, wm)
    random.setstate(rstate)
    orders = orc.shuffle_orders(3)
    ost = {k: v.clone() for k, v in full.items()}
    opt = {}
    res = orc.train_step(ost, opt, x2, y2, masks5, orders, lr=1e-4)
    close(sum(res["d_losses"]), stats["d_loss"], 1e-5, "step d_loss")
    for k, v in res["d_details"].items():
        close(v, stats[k], 2e-4 if "consist" in k else 1e-5, "step " + k, atol=1e-7)  # stats are round(.,7)
    close(res["g_loss"], stats["g_loss"], 1e-5, "step g_loss")
    for k, v in res["g_details"].items():
        close(v, stats[k], 1e-5, "step " + k, atol=1e-7)
    post = model.state_dict()
    worst = 0.0
    for k in post:
        d = (ost[k] - post[k]).abs().max().item() / (post[k].abs().max().item() + 1e-30)
        worst = max(worst, d)
    print(f"  post-step state (326 tensors) worst rel {worst:.3e}")
    assert worst < 1e-5
    # parameter *updates* (delta) are what the optimizer produced: compare those too (lr-scale)
    wd = 0.0
    for k in post:
        if k.endswith(("weight_u", "weight_v")):
            continue
        dr = post[k] - full[k]
        do = ost[k] - full[k]
        if dr.abs().max() > 0:
            wd = max(wd, (dr - do).abs().max().item() / dr.abs().max().item())
    print(f"  parameter deltas worst rel {wd:.3e}")
    assert wd < 5e-3
    assert torch.equal(post["Discriminator.c_fc.weight_orig"], full["Discriminator.c_fc.weight_orig"]), "c_fc must stay frozen (quirk 1)"
    samp = {}
    for k in sorted(post):
        idx = sample_idx(post[k].numel(), 4)
        samp[k] = [post[k].reshape(-1)[i].item() for i in idx]
    with open(os.path.join(GOLD, "step_seeded.json"), "w") as f:
        json.dump(dict(stats=stats, orders=orders, gram=res["gram"], pc_weights=res["pc_weights"],
                       d_losses=res["d_losses"].tolist(), d_grad_norms=res["d_grad_norms"], g_grad_norms=res["g_grad_norms"],
                       post_samples=samp, lr=1e-4, mask_seed=33, gfill=7, dfill=9, batch=2, data_seed=1234), f)

    # ------------------------------------------------------------------ 6. config-1 plumbing anchor (default init, 16 patches)
    print("config-1 anchor (reference default init, seeds 2024, 16 patches)")
    torch.manual_seed(2024)
    np.random.seed(2024)
    random.seed(2024)
    tape = []
    with rng_tape.recording(tape) as draws:
        m = MTD_GAN_Method()
    init_sd = {k: v.clone() for k, v in m.state_dict().items()}
    derivation = rng_tape.derive(init_sd, draws)
    del draws
    m.Discriminator.c_drop = RecDrop(0.3)
    wm = WeightMethods(method="pcgrad", n_tasks=3, device=torch.device("cpu"))
    oD = torch.optim.AdamW([dict(params=m.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4),
                            dict(params=wm.parameters(), lr=0.025, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)])
    oG = torch.optim.AdamW(m.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    loader = [dict(n_20=torch.rand(16, 1, 64, 64), n_100=torch.rand(16, 1, 64, 64))]
    rstate = random.getstate()
    stats = engine.train_MTD_GAN_Ours(m, loader, oG, oD, torch.device("cpu"), 0, 1, 2, wm)
    known = {"d_loss": 2.9324224, "g_loss": 34.6766777, "G/pix_loss": 17.0065117, "G/edge_loss": 15.6709967,
             "D/rec_loss_real": 0.4990235, "D/rec_loss_fake": 0.4334564}
    for k, v in known.items():
        close(stats[k], v, 1e-5, "BASELINE.md known answer " + k)
    random.setstate(rstate)
    orders = orc.shuffle_orders(3)
    ost = {k: v.clone() for k, v in init_sd.items()}
    res = orc.train_step(ost, {}, loader[0]["n_20"], loader[0]["n_100"], m.Discriminator.c_drop.rec, orders, lr=1e-4)
    close(sum(res["d_losses"]), stats["d_loss"], 1e-5, "anchor d_loss")
    close(res["g_loss"], stats["g_loss"], 1e-5, "anchor g_loss")
    for k, v in {**res["d_details"], **res["g_details"]}.items():
        if stats[k] > 1e-6:
            close(v, stats[k], 1e-4, "anchor " + k, atol=1e-7)
    with open(os.path.join(GOLD, "config1_anchor.json"), "w") as f:
        json.dump(dict(stats=stats, known=known), f)
    # ---- 6b. what a machine without the reference needs to REPLAY this run (tests/test_step_gpu.py): the construction's
    # random draws as a tape (oracle/rng_tape.py), float64 checksums of the 326 initial tensors and of the two input
    # batches, the five dropout masks the step drew (bit-packed keep flags), the PCGrad orders.
    torch.manual_seed(2024)
    re_sd = rng_tape.replay(tape, derivation)
    for k, v in init_sd.items():
        assert torch.equal(re_sd[k], v), "tape replay != reference construction: " + k
    rx, ry = torch.rand(16, 1, 64, 64), torch.rand(16, 1, 64, 64)
    assert torch.equal(rx, loader[0]["n_20"]) and torch.equal(ry, loader[0]["n_100"])
    masks_rec = m.Discriminator.c_drop.rec
    assert len(masks_rec) == 5 and all(tuple(k.shape) == (16, 512) for k in masks_rec)
    keep = np.stack([(k > 0).numpy() for k in masks_rec])
    assert all(torch.equal(k, torch.from_numpy(kp).float() / 0.7) for k, kp in zip(masks_rec, keep))
    np.savez_compressed(os.path.join(GOLD, "config1_replay_masks.npz"), keep_bits=np.packbits(keep.reshape(-1)))
    with open(os.path.join(GOLD, "config1_replay.json"), "w") as f:
        json.dump(dict(seed=2024, tape=tape, derivation=derivation, checksums={k: rng_tape.checksum(v) for k, v in init_sd.items()},
                       x_checksum=rng_tape.checksum(rx), y_checksum=rng_tape.checksum(ry), orders=orders, stats=stats,
                       lr=1e-4, patches=16, batch_size_arg=2), f)

    # ------------------------------------------------------------------ 7. module/pcgrad.py::PCGrad (the optimizer wrapper, a13b)
    print("module/pcgrad.py PCGrad wrapper")
    from module import pcgrad as ref_pcgrad
    import torch.nn as nn
    import torch.optim as optim
    cases = []

    def run_case(tag, net, objectives_fn, rseed, x, ys):
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        pc = ref_pcgrad.PCGrad(optim.Adam(net.parameters()))
        pc.zero_grad()
        random.seed(rseed)
        pc.pc_backward(objectives_fn(net, x, ys))
        ref_grads = [p.grad.clone() for p in net.parameters()]
        # the oracle restatement on the same flat vectors (its own backward passes, same shuffle stream)
        params = list(net.parameters())
        flat, has = [], []
        for obj in objectives_fn(net, x, ys):
            gs = torch.autograd.grad(obj, params, retain_graph=True, allow_unused=True)
            flat.append(torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]))
            has.append(torch.cat([(torch.ones_like(p) if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, params)]))
        random.seed(rseed)
        merged = orc.pcgrad_wrapper_merge(flat, has)
        close(merged, torch.cat([g.reshape(-1) for g in ref_grads]), 1e-6, f"wrapper {tag}: oracle == reference")
        cases.append(dict(tag=tag, rseed=rseed, state={k: v.tolist() for k, v in sd0.items()}, x=x.tolist(), ys=[y.tolist() for y in ys],
                          grads={n: g.tolist() for (n, _), g in zip(net.named_parameters(), ref_grads)}))

    # (a) + (b): the reference file's own self-test nets and losses (pcgrad.py:144-195), torch.manual_seed(4)
    torch.manual_seed(4)
    x, y = torch.randn(2, 3), torch.randn(2, 4)
    run_case("TestNet", ref_pcgrad.TestNet(), lambda net, x, ys: [nn.L1Loss()(net(x), ys[0]), nn.MSELoss()(net(x), ys[0])], 1, x, [y])
    torch.manual_seed(4)
    x, y = torch.randn(2, 3), torch.randn(2, 4)
    run_case("MultiHeadTestNet", ref_pcgrad.MultiHeadTestNet(),
             lambda net, x, ys: [nn.MSELoss()(net(x)[0], ys[0]), nn.MSELoss()(net(x)[1], ys[0])], 2, x, [y])
    # (c) three objectives, two of them in conflict on the shared trunk and on head 1, the third alone on head 2
    torch.manual_seed(5)
    x, y1, y2 = torch.randn(8, 3), torch.randn(8, 4), torch.randn(8, 4)
    run_case("MultiHeadTestNet/3 objectives", ref_pcgrad.MultiHeadTestNet(),
             lambda net, x, ys: [nn.MSELoss()(net(x)[0], ys[0]), nn.MSELoss()(net(x)[0], -ys[0] + 0.1), nn.L1Loss()(net(x)[1], ys[1])], 3, x, [y1, y2])
    with open(os.path.join(GOLD, "pcgrad_wrapper.json"), "w") as f:
        json.dump(cases, f)

    # ------------------------------------------------------------------ 8. Full step with seeded fill at BASELINE size (B=32)
    print("full step (seeded fill, B=32, lr 1e-4) + generator PSNR after the step")
    gold32 = {}
    for B in (2, 32):
        x32, y32 = orc.synthetic_ldct(B, seed=1234)
        model = MTD_GAN_Method()
        model.load_state_dict(full)
        masksB = mask_seq(5, B, seed=33)
        model.Discriminator.c_drop = RecDrop(0.3, inject=masksB)
        wmB = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cpu"))
        oD = torch.optim.AdamW([dict(params=model.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4),
                                dict(params=wmB.parameters(), lr=0.025, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)])
        oG = torch.optim.AdamW(model.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        random.seed(77)
        rstate = random.getstate()
        statsB = engine.train_MTD_GAN_Ours(model, [dict(n_20=x32, n_100=y32)], oG, oD, torch.device("cpu"), 0, 1, B, wmB)
        random.setstate(rstate)
        ordersB = orc.shuffle_orders(3)
        with torch.no_grad():
            pred = model.Generator(x32)                      # the generator AFTER its update (SURVEY 8d config 3 acceptance)
        post_metrics = dict(psnr=float(metrics.compute_PSNR(x32, y32, pred.clip(0, 1))[2]), ssim=float(metrics.compute_SSIM(x32, y32, pred.clip(0, 1))[2]),
                            rmse=float(metrics.compute_RMSE(x32, y32, pred.clip(0, 1))[2]))
        if B == 2:
            gold32["b2_post_metrics"] = post_metrics
            gold32["b2_post_pred_sample"] = pred[:, 0, ::8, ::8].tolist()
            continue
        ost = {k: v.clone() for k, v in full.items()}
        res = orc.train_step(ost, {}, x32, y32, masksB, ordersB, lr=1e-4)
        close(sum(res["d_losses"]), statsB["d_loss"], 1e-5, "B=32 step d_loss")
        close(res["g_loss"], statsB["g_loss"], 1e-5, "B=32 step g_loss")
        post = model.state_dict()
        worst = max((ost[k] - post[k]).abs().max().item() / (post[k].abs().max().item() + 1e-30) for k in post)
        print(f"  B=32 post-step state (326 tensors) oracle vs reference worst rel {worst:.3e}")
        assert worst < 1e-5
        close(orc.psnr(orc.generator_forward(ost, x32, "Generator.").clip(0, 1), y32), post_metrics["psnr"], 1e-6, "B=32 post-step PSNR (oracle)")
        samp = {k: [post[k].reshape(-1)[i].item() for i in sample_idx(post[k].numel(), 4)] for k in sorted(post)}
        gold32.update(stats=statsB, orders=ordersB, gram=res["gram"], pc_weights=res["pc_weights"], d_losses=res["d_losses"].tolist(),
                      d_grad_norms=res["d_grad_norms"], g_grad_norms=res["g_grad_norms"], post_samples=samp, post_metrics=post_metrics,
                      post_pred_sample=pred[:, 0, ::16, ::16].tolist(), lr=1e-4, mask_seed=33, gfill=7, dfill=9, batch=32, data_seed=1234)
    with open(os.path.join(GOLD, "step_seeded_b32.json"), "w") as f:
        json.dump(gold32, f)
    # ------------------------------------------------------------------ 9. ablation family (networks.py:478-1937)
    print("ablation family: RED-CNN generator, partial discriminators, ten wrappers (B=2, seeded fill)")
    import contextlib
    import io
    import arch.Ours.networks as ref_nets
    xa, ya = orc.synthetic_ldct(2, seed=1234)
    gold_abl = {}
    for name in orc.ABLATIONS:
        ref_m = getattr(ref_nets, name)()
        gsh = {k: tuple(v.shape) for k, v in ref_m.Generator.state_dict().items()}
        dsh = {k: tuple(v.shape) for k, v in ref_m.Discriminator.state_dict().items()}
        gst, dst = orc.seeded_fill(gsh, seed=7), orc.seeded_fill(dsh, seed=9)
        ref_m.Generator.load_state_dict(gst)
        ref_m.Discriminator.load_state_dict(dst)
        ref_m.train()
        masks_a = mask_seq(5, 2, seed=41)
        if hasattr(ref_m.Discriminator, "c_drop"):
            ref_m.Discriminator.c_drop = RecDrop(0.3, inject=[k.clone() for k in masks_a])
        full_a = {"Generator." + k: v.clone() for k, v in gst.items()}
        full_a.update({"Discriminator." + k: v.clone() for k, v in dst.items()})
        dnames = [n for n, _ in ref_m.Discriminator.named_parameters()]
        gnames = [n for n, _ in ref_m.Generator.named_parameters()]
        ost = {k: v.clone() for k, v in full_a.items()}
        for n in dnames:
            ost["Discriminator." + n] = ost["Discriminator." + n].requires_grad_(True)
        npass = 4 if orc.ABLATIONS[name][5] else 2
        with contextlib.redirect_stdout(io.StringIO()):
            d_tot, d_det = ref_m.d_loss(xa, ya)
        d_tot.backward()
        o_tot, o_det = orc.ablation_losses(name, ost, xa, ya, masks_a[:npass], "d")
        close(o_tot, d_tot, 1e-6, f"{name} d_loss")
        assert list(o_det.keys()) == list(d_det.keys()), (name, list(o_det.keys()), list(d_det.keys()))
        for k in d_det:
            close(o_det[k], d_det[k], 1e-5, f"{name} {k}", atol=1e-9)
        leaves = [ost["Discriminator." + n] for n in dnames]
        og = torch.autograd.grad(o_tot, leaves, allow_unused=True)
        dgn, worst = {}, 0.0
        for n, p_, g_ in zip(dnames, ref_m.Discriminator.parameters(), og):
            assert (p_.grad is None) == (g_ is None), (name, n)
            if p_.grad is not None:
                worst = max(worst, (g_ - p_.grad).abs().max().item() / (p_.grad.abs().max().item() + 1e-30))
                dgn[n] = p_.grad.double().norm().item()
        print(f"  {name}: D grads worst rel {worst:.2e} ({len(dgn)} of {len(dnames)} parameters reached)")
        assert worst < 1e-4
        # generator step on the same (u, v advanced) state
        for p_ in ref_m.parameters():
            p_.grad = None
        for k_ in list(ost.keys()):
            ost[k_] = ost[k_].detach()
        for n in gnames:
            ost["Generator." + n] = ost["Generator." + n].requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()):
            g_tot, g_det = ref_m.g_loss(xa, ya)
        g_tot.backward()
        o_tot, o_det = orc.ablation_losses(name, ost, xa, ya, masks_a[npass:npass + 1], "g")
        close(o_tot, g_tot, 1e-6, f"{name} g_loss")
        assert list(o_det.keys()) == list(g_det.keys()), (name, list(o_det.keys()), list(g_det.keys()))
        og = torch.autograd.grad(o_tot, [ost["Generator." + n] for n in gnames])
        ggn, worst = {}, 0.0
        for n, p_, g_ in zip(gnames, ref_m.Generator.parameters(), og):
            worst = max(worst, (g_ - p_.grad).abs().max().item() / (p_.grad.abs().max().item() + 1e-30))
            ggn[n] = p_.grad.double().norm().item()
        print(f"  {name}: G grads worst rel {worst:.2e}")
        assert worst < 1e-4
        gold_abl[name] = dict(g_keys=list(gsh.keys()), g_shapes=[list(v) for v in gsh.values()], d_keys=list(dsh.keys()),
                              d_shapes=[list(v) for v in dsh.values()], d_loss=float(d_tot), d_details={k: float(v) for k, v in d_det.items()},
                              g_loss=float(g_tot), g_details={k: float(v) for k, v in g_det.items()}, d_grad_norms=dgn, g_grad_norms=ggn,
                              d_no_grad=[n for n in dnames if n not in dgn], mask_seed=41, gfill=7, dfill=9, batch=2, data_seed=1234)
    with open(os.path.join(GOLD, "ablation.json"), "w") as f:
        json.dump(gold_abl, f)
    print(f"all pins OK in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()

"""Gradient ELEMENTS of the reference for the GPU parity tests (round 3; the norm-only checks could not see a permuted or
mis-scattered gradient).  TEST INFRASTRUCTURE, build container only (needs /root/reference):

    python oracle/pin_grad_samples.py

Writes
  tests/golden/grad_samples_b32.json       the D step at BASELINE size (32 patches, seeded fill, the state / batch / masks /
                                           `random` seed of step_seeded_b32.json): the reference's three per-task shared
                                           gradients (what module/weight_methods.py:432-433 hands to _project_conflicting),
                                           the merged gradient it writes into .grad (:438-439) and the task-specific
                                           gradients (:443), PER TENSOR: 6 sampled elements, the tensor's max-abs, and the
                                           error of the reference's own fp32 arithmetic against float64 on that tensor;
  tests/golden/g_grad_samples_b32.json     the G step of the same iteration (round 5): engine.train_MTD_GAN_Ours at 32 patches
                                           (engine.py:26-55), the generator's .grad after g_loss.backward() (engine.py:50-52;
                                           networks.py:1994-2009) PER PARAMETER -- 6 sampled elements, max-abs, the fp32-vs-
                                           float64 error, the tensor's norm -- evaluated from the state the G step starts in
                                           (the discriminator AFTER its AdamW update and four power iterations);
  tests/golden/ablation_grad_samples.json  the same per parameter for the ten Ablation_* wrappers' d_loss / g_loss gradients
                                           (the run of ablation.json: B = 2, seeded fill, mask seed 41).
Values: `ref32` = the REFERENCE's fp32 CPU result at the sampled positions; `f64` = the float64 evaluation by the oracle
restatement (pinned to the reference at rel 0.0 by oracle/pin_against_reference.py; asserted again here in fp32 before
anything is written); `err32` = max|ref32 - f64| / max|f64| over the WHOLE tensor.  A test accepts a HIP gradient element
within max(1e-3, 2 * err32) * maxabs of f64 -- DESIGN.md section 4's criterion without an oracle run at test time."""
import contextlib
import io
import json
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import _refboot  # noqa: E402

_refboot.boot()
import mtdgan_oracle as orc  # noqa: E402
import arch.Ours.networks as ref_nets  # noqa: E402
from arch.Ours.networks import MTD_GAN_Method  # noqa: E402
from module.weight_methods import WeightMethods  # noqa: E402
from pin_against_reference import RecDrop, mask_seq, sample_idx  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
K = 6


def entry(ref32, f64):
    """Golden record of one gradient tensor."""
    f64 = f64.detach().double()
    ref32 = ref32.detach()
    mx = f64.abs().max().item()
    idx = sample_idx(f64.numel(), K)
    return dict(f64=[f64.reshape(-1)[i].item() for i in idx], ref32=[ref32.reshape(-1)[i].item() for i in idx], maxabs=mx,
                err32=(ref32.double() - f64).abs().max().item() / (mx + 1e-30))


def d_step_b32():
    z2 = json.load(open(os.path.join(GOLD, "step_seeded.json")))
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    B = z["batch"]
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z2["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z2["dfill"]).items()})
    x, y = orc.synthetic_ldct(B, seed=z["data_seed"])
    masks = mask_seq(5, B, seed=z["mask_seed"])
    # ---- the reference, fp32: d_loss + WeightMethods('pcgrad').backward, per-task gradients captured on their way into the projection
    model = MTD_GAN_Method()
    model.load_state_dict(full)
    model.train()
    model.Discriminator.c_drop = RecDrop(0.3, inject=[k.clone() for k in masks])
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cpu"))
    cap = {}
    orig = wm.method._project_conflicting

    def project(grads):
        cap["tasks"] = [tuple(g.clone() for g in tg) for tg in grads]        # task order (the method shuffles `grads` in place later)
        return orig(grads)
    wm.method._project_conflicting = project
    D = model.Discriminator
    random.seed(77)
    t0 = time.time()
    losses, _ = model.d_loss(x, y)
    wm.backward(losses=losses, shared_parameters=list(D.shared_parameters()), task_specific_parameters=list(D.task_specific_parameters()),
                last_shared_parameters=list(D.last_shared_parameters()))
    print(f"  reference D step at B={B}: {time.time() - t0:.1f}s, losses {losses.tolist()}")
    names = {id(p): n for n, p in D.named_parameters()}
    shared = [names[id(p)] for p in D.shared_parameters()]
    tspec = [names[id(p)] for p in D.task_specific_parameters()]
    assert shared == orc.d_shared_names() and tspec == orc.d_task_specific_names()
    merged_ref = [p.grad.clone() for p in D.shared_parameters()]
    ts_ref = [p.grad.clone() for p in D.task_specific_parameters()]

    # ---- the oracle restatement: fp32 (must equal the reference) and float64 (the arbiter)
    def oracle(dtype):
        st = {k: v.to(dtype).clone() for k, v in full.items()}
        for n in shared + tspec:
            st["Discriminator." + n] = st["Discriminator." + n].requires_grad_(True)
        lo, _, _ = orc.d_loss(st, x.to(dtype), y.to(dtype), [k.to(dtype) for k in masks[:4]])
        sp = [st["Discriminator." + n] for n in shared]
        per_task = [torch.autograd.grad(lo[i], sp, retain_graph=True) for i in range(3)]
        ts = torch.autograd.grad(lo.sum(), [st["Discriminator." + n] for n in tspec])
        return lo.detach(), per_task, ts
    t0 = time.time()
    lo32, g32, ts32 = oracle(torch.float32)
    worst = 0.0
    for i in range(3):
        for a, b in zip(g32[i], cap["tasks"][i]):
            worst = max(worst, (a - b).abs().max().item() / (b.abs().max().item() + 1e-30))
    print(f"  oracle fp32 per-task gradients vs reference: worst rel {worst:.2e} ({time.time() - t0:.1f}s)")
    assert worst < 1e-5
    t0 = time.time()
    lo64, g64, ts64 = oracle(torch.float64)
    print(f"  oracle float64: {time.time() - t0:.1f}s")
    flat = [torch.cat([g.reshape(-1) for g in tg]) for tg in g64]
    gram = [[float(torch.dot(a, b)) for b in flat] for a in flat]
    w = orc.pcgrad_coefficients(gram, z["orders"])
    out = dict(batch=B, samples=K, orders=z["orders"], losses=lo64.tolist(), pc_weights_f64=list(map(float, w)), shared=shared, task_specific=tspec,
               tasks=[], merged={}, ts={})
    for i in range(3):
        out["tasks"].append({n: entry(r, g) for n, r, g in zip(shared, cap["tasks"][i], g64[i])})
    for j, n in enumerate(shared):
        m64 = sum(wk * tg[j] for wk, tg in zip(w, g64))
        out["merged"][n] = entry(merged_ref[j], m64)
    for n, r, g in zip(tspec, ts_ref, ts64):
        out["ts"][n] = entry(r, g)
    e = [v["err32"] for t in out["tasks"] for v in t.values()]
    print(f"  fp32-vs-float64 error of the reference per tensor: median {sorted(e)[len(e) // 2]:.1e}, max {max(e):.1e}")
    with open(os.path.join(GOLD, "grad_samples_b32.json"), "w") as f:
        json.dump(out, f)


def g_step_b32():
    """The generator gradients of the B = 32 golden iteration (tests/golden/step_seeded_b32.json).  The reference runs its own
    engine loop; the state its g_loss starts from is captured at the call, and the oracle restatement evaluates the same G step
    from that state in fp32 (must equal the reference) and in float64 (the arbiter of the GPU test)."""
    import engine as ref_engine
    z2 = json.load(open(os.path.join(GOLD, "step_seeded.json")))
    z = json.load(open(os.path.join(GOLD, "step_seeded_b32.json")))
    B = z["batch"]
    full = {"Generator." + k: v for k, v in orc.seeded_fill(orc.g_param_shapes(), seed=z2["gfill"]).items()}
    full.update({"Discriminator." + k: v for k, v in orc.seeded_fill(orc.d_state_shapes(), seed=z2["dfill"]).items()})
    x, y = orc.synthetic_ldct(B, seed=z["data_seed"])
    masks = mask_seq(5, B, seed=z["mask_seed"])
    model = MTD_GAN_Method()
    model.load_state_dict(full)
    model.Discriminator.c_drop = RecDrop(0.3, inject=[k.clone() for k in masks])
    wm = WeightMethods("pcgrad", n_tasks=3, device=torch.device("cpu"))
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    oD = torch.optim.AdamW([dict(params=model.Discriminator.parameters(), lr=z["lr"], **kw), dict(params=wm.parameters(), lr=0.025, **kw)])
    oG = torch.optim.AdamW(model.Generator.parameters(), lr=z["lr"], **kw)
    snap = {}
    real_g_loss = model.g_loss

    def g_loss(a, b):
        snap["state"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return real_g_loss(a, b)
    model.g_loss = g_loss
    random.seed(77)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        stats = ref_engine.train_MTD_GAN_Ours(model, [dict(n_20=x, n_100=y)], oG, oD, torch.device("cpu"), 0, 1, B, wm)
    print(f"  reference iteration at B={B}: {time.time() - t0:.1f}s, g_loss {stats['g_loss']}")
    assert abs(stats["g_loss"] - z["stats"]["g_loss"]) <= 1e-6 * abs(z["stats"]["g_loss"])       # the run of step_seeded_b32.json
    gnames = [n for n, _ in model.Generator.named_parameters()]
    gref = {n: p.grad.detach().clone() for n, p in model.Generator.named_parameters()}      # (optimizer.step() leaves .grad in place)

    def oracle(dtype):
        st = {k: v.to(dtype).clone() for k, v in snap["state"].items()}
        for n in gnames:
            st["Generator." + n] = st["Generator." + n].requires_grad_(True)
        total, _, _ = orc.g_loss(st, x.to(dtype), y.to(dtype), masks[4].to(dtype))
        return float(total.detach()), dict(zip(gnames, torch.autograd.grad(total, [st["Generator." + n] for n in gnames])))
    t0 = time.time()
    tot32, g32 = oracle(torch.float32)
    worst = max((g32[n] - gref[n]).abs().max().item() / (gref[n].abs().max().item() + 1e-30) for n in gnames)
    print(f"  oracle fp32 generator gradients vs reference: worst rel {worst:.2e}, g_loss {tot32} ({time.time() - t0:.1f}s)")
    assert worst < 1e-5 and abs(tot32 - stats["g_loss"]) <= 1e-5 * abs(tot32)
    t0 = time.time()
    tot64, g64 = oracle(torch.float64)
    print(f"  oracle float64: {time.time() - t0:.1f}s, g_loss {tot64}")
    out = dict(batch=B, samples=K, g_loss_f64=tot64, g_loss_ref32=stats["g_loss"], grads={})
    for n in gnames:
        e = entry(gref[n], g64[n])
        e["norm_ref32"] = float(gref[n].norm())
        e["norm_f64"] = float(g64[n].norm())
        out["grads"]["Generator." + n] = e
        assert abs(e["norm_ref32"] - z["g_grad_norms"]["Generator." + n]) <= 1e-4 * e["norm_ref32"] + 1e-12, n      # the committed norms are these
    e = [v["err32"] for v in out["grads"].values()]
    print(f"  fp32-vs-float64 error of the reference per generator tensor: median {sorted(e)[len(e) // 2]:.1e}, max {max(e):.1e}")
    with open(os.path.join(GOLD, "g_grad_samples_b32.json"), "w") as f:
        json.dump(out, f)


def ablations():
    xa, ya = orc.synthetic_ldct(2, seed=1234)
    gold = {}
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
        npass = 4 if orc.ABLATIONS[name][5] else 2
        with contextlib.redirect_stdout(io.StringIO()):
            d_tot, _ = ref_m.d_loss(xa, ya)
        d_tot.backward()
        dref = {n: p.grad.clone() for n, p in ref_m.Discriminator.named_parameters() if p.grad is not None}
        # the state after the D forward (u, v advanced) is where g_loss starts: run the oracle in both precisions along the same path
        rec = {}
        for dtype in (torch.float32, torch.float64):
            ost = {k: v.to(dtype).clone() for k, v in full_a.items()}
            for n in dnames:
                ost["Discriminator." + n] = ost["Discriminator." + n].requires_grad_(True)
            o_tot, _ = orc.ablation_losses(name, ost, xa.to(dtype), ya.to(dtype), [k.to(dtype) for k in masks_a[:npass]], "d")
            og = torch.autograd.grad(o_tot, [ost["Discriminator." + n] for n in dnames], allow_unused=True)
            rec[("d", dtype)] = {n: g for n, g in zip(dnames, og) if g is not None}
            for k_ in list(ost.keys()):
                ost[k_] = ost[k_].detach()
            for n in gnames:
                ost["Generator." + n] = ost["Generator." + n].requires_grad_(True)
            o_tot, _ = orc.ablation_losses(name, ost, xa.to(dtype), ya.to(dtype), [k.to(dtype) for k in masks_a[npass:npass + 1]], "g")
            og = torch.autograd.grad(o_tot, [ost["Generator." + n] for n in gnames])
            rec[("g", dtype)] = dict(zip(gnames, og))
        for p_ in ref_m.parameters():
            p_.grad = None
        with contextlib.redirect_stdout(io.StringIO()):
            g_tot, _ = ref_m.g_loss(xa, ya)
        g_tot.backward()
        gref = {n: p.grad.clone() for n, p in ref_m.Generator.named_parameters()}
        worst = 0.0
        for n, r in list(dref.items()) + list(gref.items()):
            o = rec[("d", torch.float32)].get(n) if n in dref and r is dref[n] else rec[("g", torch.float32)][n]
            worst = max(worst, (o - r).abs().max().item() / (r.abs().max().item() + 1e-30))
        assert worst < 1e-4, (name, worst)
        gold[name] = dict(d={n: entry(dref[n], rec[("d", torch.float64)][n]) for n in dref},
                          g={n: entry(gref[n], rec[("g", torch.float64)][n]) for n in gref})
        e = [v["err32"] for v in gold[name]["d"].values()] + [v["err32"] for v in gold[name]["g"].values()]
        print(f"  {name}: oracle fp32 == reference (worst rel {worst:.1e}); reference fp32 vs float64 per tensor: max {max(e):.1e}")
    with open(os.path.join(GOLD, "ablation_grad_samples.json"), "w") as f:
        json.dump(gold, f)


if __name__ == "__main__":
    t0 = time.time()
    if "--g-only" in sys.argv:
        g_step_b32()
        sys.exit(0)
    print("gradient elements of the ablation family")
    ablations()
    print("gradient elements of the D step at B = 32")
    d_step_b32()
    print("gradient elements of the G step at B = 32")
    g_step_b32()
    print(f"done in {time.time() - t0:.1f}s")

"""CPU oracle for the MTD-GAN generator + multi-task-discriminator training step.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the
reference algorithm, written functionally over a flat ``state`` dict (name -> tensor, the reference's
own state_dict keys).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the shipped package (mtd-gan_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED.  oracle/pin_against_reference.py imports the real reference from
/root/reference in the build container, checks every function below against it on seeded inputs
(fp32, tolerances in that script) and writes tests/golden/*.npz.  tests/test_oracle_golden.py
re-checks the oracle against those committed vectors on any machine.

The arithmetic itself lives in a third-party dependency of the reference, PyTorch (reference pins
torch==2.3.1 in requirements.txt:16; this container has 2.10.0): conv2d / conv_transpose2d /
fft.rfft2 / fft.irfft2 / spectral_norm / interpolate / pixel_shuffle.  Those call sites are restated
here with explicit formulas where the semantics are subtle (irfft2 on a non-Hermitian spectrum,
spectral-norm power iteration, PCGrad as a function of the Gram matrix).

Reference anchors (file:line under /root/reference):
  resfft_block            arch/Ours/networks.py:15-36
  generator_forward       arch/Ours/networks.py:95-164  (ctor args 1,32,10,3,1 from :1944)
  sn_weight               torch.nn.utils.spectral_norm as applied at arch/Ours/networks.py:181-300
  discriminator_forward   arch/Ours/networks.py:383-474
  ls_gan / nds_loss       losses.py:10-15
  charbonnier / edge_loss losses.py:99-138
  d_loss / g_loss         arch/Ours/networks.py:1957-2009
  pcgrad_*                module/weight_methods.py:429-468
  pcgrad_wrapper_merge    module/pcgrad.py:50-69 (the optimizer wrapper's projection + mean / sum reduction)
  redcnn_forward          arch/Ours/networks.py:478-505
  ablation_losses         arch/Ours/networks.py:1324-1937 (the ten Ablation_* wrappers; partial discriminators :507-1322)
  adamw_step              train.py:122-126, optimizers.py:8-9 (torch.optim.AdamW)
  train_step              engine.py:33-55
  psnr / ssim / rmse      metrics.py:172-244
"""
import math
import random
from collections import OrderedDict

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# Layer tables (shape contract of the reference; used to build states and to walk the networks)
# ----------------------------------------------------------------------------------------------
G_CH = 32          # arch/Ours/networks.py:1944  out_channels=32, kernel 3, padding 1, 10 layers
G_LAYERS = 10
G_BLOCKS = 21

# (name, cin, cout, k, stride, pad) -- spectral-normalised convs of the discriminator, C = 64
_C = 64
D_TRUNK = []
_cin = 1
for _lvl, _co in enumerate([_C, _C * 2, _C * 4, _C * 8, _C * 8, _C * 8], start=1):
    D_TRUNK.append((f"conv{_lvl}1", _cin, _co, 3, 1, 1))
    D_TRUNK.append((f"conv{_lvl}2", _co, _co, 3, 1, 1))
    D_TRUNK.append((f"down{_lvl}", _co, _co, 4, 2, 1))
    _cin = _co
D_BOT = [("bconv1", _C * 8, _C * 8, 1, 1, 0), ("bconv2", _C * 8, _C * 8, 1, 1, 0)]
# decoder level k (1..6): (cin of cat, cout)
D_DEC = [(_C * 16, _C * 8), (_C * 16, _C * 8), (_C * 16, _C * 4), (_C * 8, _C * 2), (_C * 4, _C), (_C * 2, 1)]
# r_up{k}: 1x1 conv cin -> 4*cout', PixelShuffle(2)   (arch/Ours/networks.py:267-300)
D_RUP = [(_C * 8, _C * 8), (_C * 8, _C * 8), (_C * 8, _C * 8), (_C * 4, _C * 4), (_C * 2, _C * 2), (_C, _C)]


def d_sn_layers():
    """All spectral-normalised layers as (name, weight shape).  45 entries."""
    out = [(n, (co, ci, k, k)) for (n, ci, co, k, s, p) in D_TRUNK + D_BOT]
    out.append(("c_fc", (512, 512)))
    for pre in ("s", "r"):
        for lvl, (ci, co) in enumerate(D_DEC, start=1):
            out.append((f"{pre}_dconv{lvl}1", (co, ci, 3, 3)))
            out.append((f"{pre}_dconv{lvl}2", (co, co, 3, 3)))
    return out


def d_shared_names():
    """arch/Ours/networks.py:318-340 -- per layer `bias, weight_orig` (Parameter registration order)."""
    names = []
    for (n, *_r) in D_TRUNK + D_BOT:
        names += [f"{n}.bias", f"{n}.weight_orig"]
    return names


def d_task_specific_names():
    """arch/Ours/networks.py:342-377.  `c_fc` is in neither list (reference quirk)."""
    names = []
    for lvl in range(1, 7):
        names += [f"s_dconv{lvl}1.bias", f"s_dconv{lvl}1.weight_orig", f"s_dconv{lvl}2.bias", f"s_dconv{lvl}2.weight_orig"]
    for lvl in range(1, 7):
        names += [f"r_up{lvl}.upsample.0.weight", f"r_up{lvl}.upsample.0.bias"]
        names += [f"r_dconv{lvl}1.bias", f"r_dconv{lvl}1.weight_orig", f"r_dconv{lvl}2.bias", f"r_dconv{lvl}2.weight_orig"]
    names += ["enc_out.weight", "enc_out.bias", "dec_out.weight", "dec_out.bias", "rec_out.weight", "rec_out.bias"]
    return names


def g_param_shapes():
    sh = OrderedDict()
    for i in range(G_LAYERS + 1):
        sh[f"encoder.{i}.weight"] = (G_CH, 1 if i == 0 else G_CH, 3, 3)
        sh[f"encoder.{i}.bias"] = (G_CH,)
    for i in range(G_LAYERS + 1):
        # ConvTranspose2d weight layout is (Cin, Cout, kh, kw); decoder.0 is 32 -> 1
        sh[f"decoder.{i}.weight"] = (G_CH, 1 if i == 0 else G_CH, 3, 3)
        sh[f"decoder.{i}.bias"] = (1 if i == 0 else G_CH,)
    for i in range(G_BLOCKS):
        sh[f"enforce.{i}.img_conv.weight"] = (G_CH, G_CH, 3, 3)
        sh[f"enforce.{i}.img_conv.bias"] = (G_CH,)
        sh[f"enforce.{i}.fft_conv.weight"] = (2 * G_CH, 2 * G_CH, 1, 1)
        sh[f"enforce.{i}.fft_conv.bias"] = (2 * G_CH,)
    return sh


def d_state_shapes():
    """Every entry of Discriminator.state_dict() (params and the spectral-norm u/v buffers)."""
    sh = OrderedDict()
    for n, wshape in d_sn_layers():
        co = wshape[0]
        kk = 1
        for d in wshape[1:]:
            kk *= d
        sh[f"{n}.bias"] = (co,)
        sh[f"{n}.weight_orig"] = wshape
        sh[f"{n}.weight_u"] = (co,)
        sh[f"{n}.weight_v"] = (kk,)
    for lvl, (ci, co) in enumerate(D_RUP, start=1):
        sh[f"r_up{lvl}.upsample.0.weight"] = (4 * co, ci, 1, 1)
        sh[f"r_up{lvl}.upsample.0.bias"] = (4 * co,)
    sh["enc_out.weight"] = (1, 512)
    sh["enc_out.bias"] = (1,)
    sh["dec_out.weight"] = (1, 1, 1, 1)
    sh["dec_out.bias"] = (1,)
    sh["rec_out.weight"] = (1, 1, 1, 1)
    sh["rec_out.bias"] = (1,)
    return sh


# ----------------------------------------------------------------------------------------------
# Seeded fill recipe (SURVEY §8c "fixture design"): identical for the reference, the oracle and the
# HIP build.  Keys are visited in sorted order; each tensor gets its own generator seed.
# ----------------------------------------------------------------------------------------------
def seeded_fill(shapes, seed, g_gain=0.3, d_gain=1.0, dtype=torch.float32):
    state = OrderedDict()
    for idx, name in enumerate(sorted(shapes.keys())):
        shape = tuple(shapes[name])
        gen = torch.Generator().manual_seed(seed * 100003 + idx)
        t = torch.randn(shape, generator=gen, dtype=torch.float64)
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("weight_u", "weight_v"):
            t = t / t.norm().clamp_min(1e-12)
        elif leaf == "bias":
            t = t * 0.1
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            if name.startswith(("encoder.", "decoder.", "enforce.")):
                if name.startswith("decoder."):
                    fan_in = shape[0] * shape[2] * shape[3]   # ConvTranspose: (Cin, Cout, k, k)
                t = t * (g_gain / math.sqrt(fan_in))
            else:
                t = t * (d_gain / math.sqrt(fan_in))
        state[name] = t.to(dtype)
    # spectral-norm buffers: 3 power iterations from the random start (float64), so that eval-mode
    # forwards see a meaningful sigma (a random u.v pair gives sigma ~ 0 and the net overflows)
    for name in [n for n in state if n.endswith(".weight_u")]:
        base = name[: -len("weight_u")]
        wm = state[base + "weight_orig"].double().reshape(state[name].shape[0], -1)
        u = state[name].double()
        for _ in range(3):
            v = F.normalize(wm.t() @ u, dim=0, eps=1e-12)
            u = F.normalize(wm @ v, dim=0, eps=1e-12)
        state[name] = u.to(dtype)
        state[base + "weight_v"] = v.to(dtype)
    return state


def synthetic_ldct(batch, seed=1234, size=64, dtype=torch.float32):
    """LDCT-shaped synthetic patches.  Mimics create_datasets/Mayo.py:119-136:
    HU field -> clip((HU+160)/400, 0, 1).  Returns (x low-dose, y normal-dose), NCHW with C=1."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(batch, 1, size, size, generator=g)
    box = torch.full((1, 1, 5, 5), 1.0 / 25.0)
    for _ in range(3):
        z = F.conv2d(F.pad(z, (2, 2, 2, 2), mode="replicate"), box)
    hu = 150.0 * z / z.std() + 40.0
    noise = torch.randn(batch, 1, size, size, generator=g)
    y = ((hu + 160.0) / 400.0).clamp(0, 1)
    x = ((hu + 40.0 * noise + 160.0) / 400.0).clamp(0, 1)
    return x.to(dtype), y.to(dtype)


# ----------------------------------------------------------------------------------------------
# Generator
# ----------------------------------------------------------------------------------------------
def irfft2_ortho_explicit(zr, zi, H, W):
    """irfft2(complex(zr, zi), s=(H,W), norm='ortho') written out: complex inverse
    DFT along H for all W/2+1 columns, then c2r along W that uses only Re of columns 0 and W/2."""
    z = torch.complex(zr, zi)
    t = torch.fft.ifft(z, n=H, dim=-2, norm="ortho")                     # along H, all 33 columns
    kw = torch.arange(W // 2 + 1, dtype=zr.dtype)
    w = torch.arange(W, dtype=zr.dtype)
    ang = 2.0 * math.pi * kw[:, None] * w[None, :] / W                   # (33, W)
    wt = torch.full((W // 2 + 1,), 2.0, dtype=zr.dtype)
    wt[0] = 1.0
    wt[-1] = 1.0
    cosm = torch.cos(ang) * wt[:, None]
    sinm = torch.sin(ang) * wt[:, None]
    sinm[0] = 0.0
    sinm[-1] = 0.0
    return (t.real @ cosm - t.imag @ sinm) / math.sqrt(W)


def resfft_block(x, w_img, b_img, w_fft, b_fft, explicit_irfft=False):
    """arch/Ours/networks.py:21-36: x + relu(conv3x3(x)) + irfft2(relu(conv1x1([Re;Im] rfft2 x)))."""
    H, W = x.shape[-2:]
    f = torch.fft.rfft2(x, s=(H, W), dim=(2, 3), norm="ortho")
    cat = torch.cat([f.real, f.imag], dim=1)
    z = F.relu(F.conv2d(cat, w_fft, b_fft))
    zr, zi = torch.chunk(z, 2, dim=1)
    if explicit_irfft:
        y = irfft2_ortho_explicit(zr, zi, H, W)
    else:
        y = torch.fft.irfft2(torch.complex(zr, zi), s=(H, W), dim=(2, 3), norm="ortho")
    img = F.relu(F.conv2d(x, w_img, b_img, padding=1))
    return x + img + y


def _blk(state, i, x, pre="", **kw):
    p = f"{pre}enforce.{i}."
    return resfft_block(x, state[p + "img_conv.weight"], state[p + "img_conv.bias"],
                        state[p + "fft_conv.weight"], state[p + "fft_conv.bias"], **kw)


def generator_forward(state, x, pre="", **kw):
    """arch/Ours/networks.py:95-164.  `pre` is the key prefix ('' or 'Generator.')."""
    enc = lambda i, t: F.relu(F.conv2d(t, state[f"{pre}encoder.{i}.weight"], state[f"{pre}encoder.{i}.bias"], padding=1))
    dec = lambda i, t: F.conv_transpose2d(t, state[f"{pre}decoder.{i}.weight"], state[f"{pre}decoder.{i}.bias"], padding=1)
    skips = []
    t = x
    for i in range(G_LAYERS):                      # e1..e10
        t = _blk(state, i, enc(i, t), pre, **kw)
        skips.append(t)
    t = _blk(state, G_LAYERS, enc(G_LAYERS, t), pre, **kw)          # bottleneck
    t = F.relu(dec(G_LAYERS, t) + skips[G_LAYERS - 1])               # d10 = relu(decoder[-1](x_b) + e10)
    for j in range(1, G_LAYERS):                   # d9..d1
        t = _blk(state, G_LAYERS + j, t, pre, **kw)
        t = F.relu(dec(G_LAYERS - j, t) + skips[G_LAYERS - 1 - j])
    t = _blk(state, 2 * G_LAYERS, t, pre, **kw)
    return F.relu(dec(0, t) + x)


# ----------------------------------------------------------------------------------------------
# Spectral norm (old hook API, n_power_iterations=1, eps=1e-12, dim=0)
# ----------------------------------------------------------------------------------------------
def sn_weight(w_orig, u, v, train, eps=1e-12):
    """Returns (W/sigma, u_new, v_new).  In train mode u, v are updated first (no grad), then
    sigma = u . (W_mat v) with u, v treated as constants, so grad flows through sigma into W_orig."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


class DState:
    """Mutable view over a discriminator state (so that successive forwards see updated u/v)."""

    def __init__(self, state, pre=""):
        self.s = state
        self.pre = pre

    def sn(self, name, train):
        p = self.pre + name
        w, u, v = sn_weight(self.s[p + ".weight_orig"], self.s[p + ".weight_u"], self.s[p + ".weight_v"], train)
        if train:
            self.s[p + ".weight_u"] = u
            self.s[p + ".weight_v"] = v
        return w, self.s[p + ".bias"]

    def __getitem__(self, k):
        return self.s[self.pre + k]


def discriminator_forward(state, x, train=True, drop_mask=None, pre="", need_rec=True, heads=("cls", "seg", "rec"), seg_prefix="s_"):
    """arch/Ours/networks.py:383-474.  Mutates the u/v entries of `state` when train=True, exactly
    as the reference's forward pre-hook does.  drop_mask: (B,512) multiplier (0 or 1/(1-p)); None =
    no dropout (eval, or p=0).  Module call order (and so the u/v update order) follows forward().
    heads / seg_prefix: the ablation discriminators (networks.py:507-1322) are this network with a subset of the heads
    (outputs of absent heads are None); SEG_Discriminator names its decoder layers without the 's_' prefix."""
    st = DState(state, pre)
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    t = x
    skips = []
    for lvl in range(1, 7):
        w, b = st.sn(f"conv{lvl}1", train)
        t = lrelu(F.conv2d(t, w, b, padding=1))
        w, b = st.sn(f"conv{lvl}2", train)
        t = lrelu(F.conv2d(t, w, b, padding=1))
        skips.append(t)
        w, b = st.sn(f"down{lvl}", train)
        t = F.conv2d(t, w, b, stride=2, padding=1)          # no activation after the strided conv
    w, b = st.sn("bconv1", train)
    t = lrelu(F.conv2d(t, w, b))
    w, b = st.sn("bconv2", train)
    bot = lrelu(F.conv2d(t, w, b))
    # CLS
    c = None
    if "cls" in heads:
        w, b = st.sn("c_fc", train)
        c = lrelu(F.linear(bot.flatten(1), w, b))
        if drop_mask is not None:
            c = c * drop_mask
    # SEG
    t = bot
    for lvl in range(1, 7) if "seg" in heads else ():
        t = F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
        w, b = st.sn(f"{seg_prefix}dconv{lvl}1", train)
        t = lrelu(F.conv2d(torch.cat([t, skips[6 - lvl]], dim=1), w, b, padding=1))
        w, b = st.sn(f"{seg_prefix}dconv{lvl}2", train)
        t = lrelu(F.conv2d(t, w, b, padding=1))
    seg = t
    # REC
    rec = None
    if need_rec and "rec" in heads:
        t = bot
        for lvl in range(1, 7):
            t = F.pixel_shuffle(F.conv2d(t, st[f"r_up{lvl}.upsample.0.weight"], st[f"r_up{lvl}.upsample.0.bias"]), 2)
            w, b = st.sn(f"r_dconv{lvl}1", train)
            t = lrelu(F.conv2d(torch.cat([t, skips[6 - lvl]], dim=1), w, b, padding=1))
            w, b = st.sn(f"r_dconv{lvl}2", train)
            t = lrelu(F.conv2d(t, w, b, padding=1))
        rec = F.conv2d(t, st["rec_out.weight"], st["rec_out.bias"])
    x_enc = F.linear(c, st["enc_out.weight"], st["enc_out.bias"]) if "cls" in heads else None
    x_dec = F.conv2d(seg, st["dec_out.weight"], st["dec_out.bias"]) if "seg" in heads else None
    return x_enc, x_dec, rec


# ----------------------------------------------------------------------------------------------
# Ablation family (arch/Ours/networks.py:478-1937)
# ----------------------------------------------------------------------------------------------
def redcnn_forward(state, x, pre=""):
    """arch/Ours/networks.py:498-505 (REDCNN_Generator with 1,32,10,3,1)."""
    residuals = []
    t = x
    for i in range(G_LAYERS + 1):
        residuals.append(t)
        t = F.relu(F.conv2d(t, state[f"{pre}encoder.{i}.weight"], state[f"{pre}encoder.{i}.bias"], padding=1))
    for i in range(G_LAYERS, -1, -1):
        t = F.relu(F.conv_transpose2d(t, state[f"{pre}decoder.{i}.weight"], state[f"{pre}decoder.{i}.bias"], padding=1) + residuals[i])
    return t


# name -> (generator, discriminator heads, seg prefix, discriminator outputs in order, NDS, RC, terms of the generator's
# adversarial loss as (output, logged name), 'real_enc + real_dec + fake_enc + fake_dec' summation order)
ABLATIONS = {
    "Ablation_CLS": ("redcnn", ("cls",), "s_", ("enc",), False, False, (("enc", "G/gen_enc"),), False),
    "Ablation_SEG": ("redcnn", ("seg",), "", ("enc",), False, False, (("enc", "G/gen_enc"),), False),
    "Ablation_CLS_SEG": ("redcnn", ("cls", "seg"), "s_", ("enc", "dec"), False, False, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), True),
    "Ablation_CLS_REC": ("redcnn", ("cls", "rec"), "s_", ("enc", "rec"), False, False, (("enc", "G/gen_enc"), ("rec", "G/gen_dec")), False),
    "Ablation_SEG_REC": ("redcnn", ("seg", "rec"), "s_", ("dec", "rec"), False, False, (("dec", "G/gen_enc"), ("rec", "G/gen_dec")), False),
    "Ablation_CLS_SEG_REC": ("redcnn", ("cls", "seg", "rec"), "s_", ("enc", "dec", "rec"), False, False, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), True),
    "Ablation_CLS_SEG_REC_NDS": ("redcnn", ("cls", "seg", "rec"), "s_", ("enc", "dec", "rec"), True, False, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), False),
    "Ablation_CLS_SEG_REC_RC": ("redcnn", ("cls", "seg", "rec"), "s_", ("enc", "dec", "rec"), False, True, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), True),
    "Ablation_CLS_SEG_REC_NDS_RC": ("redcnn", ("cls", "seg", "rec"), "s_", ("enc", "dec", "rec"), True, True, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), False),
    "Ablation_CLS_SEG_REC_NDS_RC_ResFFT": ("resfft", ("cls", "seg", "rec"), "s_", ("enc", "dec", "rec"), True, True, (("enc", "G/gen_enc"), ("dec", "G/gen_dec")), False),
}


def ablation_losses(name, state, x, y, drop_masks, which, gpre="Generator.", dpre="Discriminator."):
    """d_loss (which='d') or g_loss (which='g') of one of the ten ablation wrappers (networks.py:1324-1937), restated.
    drop_masks: one (B,512) multiplier per discriminator pass of the call, in order (ignored by heads without the image-level
    branch).  Returns (total, details).  Mutates the spectral-norm u/v of `state` like the reference."""
    gen, heads, spre, outs, nds, rc, gterms, interleaved = ABLATIONS[name]
    masks = list(drop_masks)
    G = (lambda t: redcnn_forward(state, t, gpre)) if gen == "redcnn" else (lambda t: generator_forward(state, t, gpre))

    def D(t):
        m = masks.pop(0) if masks else None
        e, d, r = discriminator_forward(state, t, True, m if "cls" in heads else None, dpre, True, heads, spre)
        vals = [v for v in (e, d, r) if v is not None]
        return dict(zip(outs, vals))
    seg = (lambda t, tgt: nds_loss(t, tgt, x - y)) if nds else ls_gan
    details = OrderedDict()
    if which == "d":
        with torch.no_grad():
            fake = G(x)
        real, fk = D(y), D(fake)
        if "enc" in real:
            details["D/real_enc"], details["D/fake_enc"] = ls_gan(real["enc"], 1.0), ls_gan(fk["enc"], 0.0)
        if "dec" in real:
            details["D/real_dec"], details["D/fake_dec"] = seg(real["dec"], 1.0), seg(fk["dec"], 0.0)
        order = ["D/real_enc", "D/real_dec", "D/fake_enc", "D/fake_dec"] if interleaved else ["D/real_enc", "D/fake_enc", "D/real_dec", "D/fake_dec"]
        terms = [details[k] for k in order if k in details]
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        if "rec" in real:
            details["D/rec_loss_real"] = F.l1_loss(real["rec"], y)
            details["D/rec_loss_fake"] = F.l1_loss(fk["rec"], fake)
            total = total + (details["D/rec_loss_real"] + details["D/rec_loss_fake"])
        if rc:
            rr, rf = D(real["rec"].clip(0, 1)), D(fk["rec"].clip(0, 1))
            c1, c2 = F.mse_loss(real["enc"], rr["enc"]), F.mse_loss(real["dec"], rr["dec"])
            c3, c4 = F.mse_loss(fk["enc"], rf["enc"]), F.mse_loss(fk["dec"], rf["dec"])
            details["D/consist_loss_real_enc"], details["D/consist_loss_real_dec"] = c1, c2
            details["D/consist_loss_fake_enc"], details["D/consist_loss_fake_dec"] = c3, c4
            total = total + (c1 + c2 + c3 + c4)
        return total, details
    fake = G(x)
    gen_out = D(fake)
    adv = None
    for key, nm in gterms:
        t = seg(gen_out[key], 1.0) if (key == "dec" and nm == "G/gen_dec") else ls_gan(gen_out[key], 1.0)
        details[nm] = t
        adv = t if adv is None else adv + t
    pix = 50.0 * charbonnier(fake, y)
    edge = 50.0 * edge_loss(fake, y)
    details["G/pix_loss"], details["G/edge_loss"] = pix, edge
    return adv + pix + edge, details


# ----------------------------------------------------------------------------------------------
# Losses
# ----------------------------------------------------------------------------------------------
def ls_gan(a, target):
    return torch.mean((a - target) ** 2)


def nds_loss(a, target, diffs):
    """losses.py:13-15: mean over ALL elements of bool(|diffs|) * (a-target)^2."""
    return torch.mean((diffs != 0).to(a.dtype) * (a - target) ** 2)


def charbonnier(a, b, eps=1e-3):
    d = a - b
    return torch.mean(torch.sqrt(d * d + eps * eps))


_GAUSS_1D = (0.05, 0.25, 0.4, 0.25, 0.05)


def _gauss5(img):
    k1 = torch.tensor(_GAUSS_1D, dtype=img.dtype)
    k = torch.outer(k1, k1)[None, None]
    return F.conv2d(F.pad(img, (2, 2, 2, 2), mode="replicate"), k)


def laplacian(img):
    """losses.py:125-132: img - gauss(zero-insert-upsample(4 * gauss(img)[::2, ::2]))."""
    f = _gauss5(img)
    up = torch.zeros_like(f)
    up[:, :, ::2, ::2] = f[:, :, ::2, ::2] * 4
    return img - _gauss5(up)


def edge_loss(a, b):
    return charbonnier(laplacian(a), laplacian(b))


def d_loss(state, x, y, drop_masks=(None, None, None, None), train=True, gpre="Generator.", dpre="Discriminator."):
    """arch/Ours/networks.py:1957-1992.  Returns (stack[disc, rec, consist], details, fake)."""
    with torch.no_grad():
        fake = generator_forward(state, x, gpre)
    diff = x - y
    real_enc, real_dec, real_rec = discriminator_forward(state, y, train, drop_masks[0], dpre)
    fake_enc, fake_dec, fake_rec = discriminator_forward(state, fake, train, drop_masks[1], dpre)
    t_real_enc = ls_gan(real_enc, 1.0)
    t_fake_enc = ls_gan(fake_enc, 0.0)
    t_real_dec = nds_loss(real_dec, 1.0, diff)
    t_fake_dec = nds_loss(fake_dec, 0.0, diff)
    disc = t_real_enc + t_fake_enc + t_real_dec + t_fake_dec
    rec_real = F.l1_loss(real_rec, y)
    rec_fake = F.l1_loss(fake_rec, fake)
    rec = rec_real + rec_fake
    rr_enc, rr_dec, _ = discriminator_forward(state, real_rec.clip(0, 1), train, drop_masks[2], dpre)
    rf_enc, rf_dec, _ = discriminator_forward(state, fake_rec.clip(0, 1), train, drop_masks[3], dpre)
    c1 = F.mse_loss(real_enc, rr_enc)
    c2 = F.mse_loss(real_dec, rr_dec)
    c3 = F.mse_loss(fake_enc, rf_enc)
    c4 = F.mse_loss(fake_dec, rf_dec)
    consist = c1 + c2 + c3 + c4
    details = OrderedDict([
        ("D/real_enc", t_real_enc), ("D/fake_enc", t_fake_enc), ("D/real_dec", t_real_dec), ("D/fake_dec", t_fake_dec),
        ("D/rec_loss_real", rec_real), ("D/rec_loss_fake", rec_fake),
        ("D/consist_loss_real_enc", c1), ("D/consist_loss_real_dec", c2),
        ("D/consist_loss_fake_enc", c3), ("D/consist_loss_fake_dec", c4)])
    return torch.stack([disc, rec, consist]), details, fake


def g_loss(state, x, y, drop_mask=None, train=True, gpre="Generator.", dpre="Discriminator."):
    """arch/Ours/networks.py:1994-2009."""
    fake = generator_forward(state, x, gpre)
    gen_enc, gen_dec, _ = discriminator_forward(state, fake, train, drop_mask, dpre)
    diff = x - y
    t_enc = ls_gan(gen_enc, 1.0)
    t_dec = nds_loss(gen_dec, 1.0, diff)
    pix = 50.0 * charbonnier(fake, y)
    edge = 50.0 * edge_loss(fake, y)
    total = t_enc + t_dec + pix + edge
    details = OrderedDict([("G/gen_enc", t_enc), ("G/gen_dec", t_dec), ("G/pix_loss", pix), ("G/edge_loss", edge)])
    return total, details, fake


# ----------------------------------------------------------------------------------------------
# PCGrad (module/weight_methods.py:449-464)
# ----------------------------------------------------------------------------------------------
def pcgrad_merge(task_grads, rng=random):
    """Literal restatement on flat vectors: task_grads is a list of T 1-D tensors.  Consumes
    rng.shuffle exactly like the reference (one in-place shuffle of the shared list per i)."""
    grads = list(task_grads)
    pc = [g.clone() for g in task_grads]
    for gi in pc:
        rng.shuffle(grads)
        for gj in grads:
            d = torch.dot(gi, gj)
            if d < 0:
                gi -= d * gj / (gj.norm() ** 2)
    return sum(pc)


def pcgrad_coefficients(gram, orders):
    """PCGrad as a function of the Gram matrix of the ORIGINAL task gradients.
    gram: (T,T) python/numpy/tensor;  orders: for each i the index order after that i's shuffle.
    Returns w with merged = sum_k w[k] * g_k."""
    T = len(orders)
    w = [0.0] * T
    for i in range(T):
        c = [0.0] * T
        c[i] = 1.0
        for j in orders[i]:
            d = sum(c[k] * float(gram[k][j]) for k in range(T))
            if d < 0:
                c[j] -= d / float(gram[j][j])
        for k in range(T):
            w[k] += c[k]
    return w


def shuffle_orders(T, rng=random):
    """Index orders produced by the reference's cumulative in-place shuffles of the grads list."""
    idx = list(range(T))
    orders = []
    for _ in range(T):
        rng.shuffle(idx)
        orders.append(list(idx))
    return orders


def pcgrad_wrapper_merge(grads, has_grads, reduction="mean", rng=random):
    """module/pcgrad.py:50-69 restated on flat vectors.  grads: T flat gradients over ALL optimizer parameters (zeros
    where an objective does not reach a parameter), has_grads: T flat 0/1 masks.  The projection runs over the whole
    vector with one cumulative in-place shuffle of the task list per i; elements every objective reaches get the mean
    of the projected gradients (`if self._reduction:` is truthy for 'mean' and 'sum' alike), the others their sum."""
    if not reduction:
        raise ValueError("invalid reduction method")
    shared = torch.stack(has_grads).prod(0).bool()
    order = list(grads)
    pc = [g.clone() for g in grads]
    for gi in pc:
        rng.shuffle(order)
        for gj in order:
            d = torch.dot(gi, gj)
            if d < 0:
                gi -= d * gj / (gj.norm() ** 2)
    merged = torch.zeros_like(grads[0])
    merged[shared] = torch.stack([g[shared] for g in pc]).mean(dim=0)
    merged[~shared] = torch.stack([g[~shared] for g in pc]).sum(dim=0)
    return merged


# ----------------------------------------------------------------------------------------------
# AdamW (torch.optim.AdamW defaults used by the reference: betas (0.9,0.999), eps 1e-8, wd 5e-4)
# ----------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=5e-4):
    """One decoupled-weight-decay Adam update; returns (p, m, v).  `step` is 1-based."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# ----------------------------------------------------------------------------------------------
# Pixel metrics (metrics.py:172-244) -- acceptance metrics, data range 1.0
# ----------------------------------------------------------------------------------------------
def psnr(pred, gt, data_range=1.0):
    """metrics.py:184-197: MSE over the WHOLE batch tensor, +1e-10."""
    mse = torch.mean((pred - gt) ** 2) + 1e-10
    return 10.0 * torch.log10(data_range ** 2 / mse)


def rmse(pred, gt):
    """metrics.py:174-181."""
    return torch.sqrt(torch.mean((pred - gt) ** 2))


def ssim(pred, gt, data_range=1.0, window_size=11, sigma=1.5):
    coords = torch.arange(window_size, dtype=pred.dtype) - window_size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    win = torch.outer(g, g)[None, None]
    pad = window_size // 2
    mu1 = F.conv2d(pred, win, padding=pad)
    mu2 = F.conv2d(gt, win, padding=pad)
    s11 = F.conv2d(pred * pred, win, padding=pad) - mu1 * mu1
    s22 = F.conv2d(gt * gt, win, padding=pad) - mu2 * mu2
    s12 = F.conv2d(pred * gt, win, padding=pad) - mu1 * mu2
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean()


# ----------------------------------------------------------------------------------------------
# One full training iteration (engine.py:33-55 with method_D = PCGrad), functional.
# ----------------------------------------------------------------------------------------------
def train_step(state, opt, x, y, drop_masks, orders, lr=1e-4, wd=5e-4):
    """state: full 'Generator.*' / 'Discriminator.*' dict (mutated in place: params, u/v).
    opt: dict name -> (m, v, step) AdamW state (mutated).  drop_masks: 5 masks (4 D-step forwards,
    1 G-step forward).  orders: PCGrad shuffle orders (3 lists).  Returns a dict of results."""
    dpre, gpre = "Discriminator.", "Generator."
    shared = [dpre + n for n in d_shared_names()]
    tspec = [dpre + n for n in d_task_specific_names()]
    # ---- D step -----------------------------------------------------------------------------
    for n in shared + tspec:
        state[n] = state[n].detach().requires_grad_(True)
    losses, details, _ = d_loss(state, x, y, drop_masks[:4], True, gpre, dpre)
    sp = [state[n] for n in shared]
    tp = [state[n] for n in tspec]
    task_grads = [torch.autograd.grad(losses[i], sp, retain_graph=True) for i in range(3)]
    ts_grads = torch.autograd.grad(losses.sum(), tp)
    flat = [torch.cat([g.reshape(-1) for g in tg]) for tg in task_grads]
    gram = [[float(torch.dot(flat[a].double(), flat[b].double())) for b in range(3)] for a in range(3)]
    wts = pcgrad_coefficients(gram, orders)
    merged = sum(w * f for w, f in zip(wts, flat))
    out = {"d_losses": losses.detach().clone(), "d_details": {k: float(v.detach()) for k, v in details.items()},
           "gram": gram, "pc_weights": wts}
    ofs = 0
    grads = {}
    for n, p in zip(shared, sp):
        grads[n] = merged[ofs:ofs + p.numel()].reshape(p.shape)
        ofs += p.numel()
    for n, g in zip(tspec, ts_grads):
        grads[n] = g
    out["d_grad_norms"] = {n: float(g.norm()) for n, g in grads.items()}
    for n, g in grads.items():
        m, v, step = opt.get(n, (torch.zeros_like(g), torch.zeros_like(g), 0))
        p, m, v = adamw_step(state[n].detach(), g, m, v, step + 1, lr, wd=wd)
        state[n] = p
        opt[n] = (m, v, step + 1)
    for n in list(state.keys()):
        state[n] = state[n].detach()
    # ---- G step -----------------------------------------------------------------------------
    gnames = [n for n in state if n.startswith(gpre)]
    for n in gnames:
        state[n] = state[n].detach().requires_grad_(True)
    total, gdetails, fake = g_loss(state, x, y, drop_masks[4], True, gpre, dpre)
    gp = [state[n] for n in gnames]
    ggrads = torch.autograd.grad(total, gp)
    out["g_loss"] = float(total.detach())
    out["g_details"] = {k: float(v.detach()) for k, v in gdetails.items()}
    out["g_grad_norms"] = {n: float(g.norm()) for n, g in zip(gnames, ggrads)}
    for n, g in zip(gnames, ggrads):
        m, v, step = opt.get(n, (torch.zeros_like(g), torch.zeros_like(g), 0))
        p, m, v = adamw_step(state[n].detach(), g, m, v, step + 1, lr, wd=wd)
        state[n] = p
        opt[n] = (m, v, step + 1)
    for n in list(state.keys()):
        state[n] = state[n].detach()
    out["fake"] = fake.detach()
    return out

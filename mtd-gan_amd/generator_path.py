"""Forward / backward schedules of the Res-FFT-Conv block and the ResFFT generator on the HIP kernels.

Mirrors arch/Ours/networks.py:21-36 (FFT_ConvBlock.forward) and :95-164 (ResFFT_Generator.forward) of
the reference; the backward schedules are their autograd transposes written out by hand so that one
`torch.autograd.Function` covers the whole generator (no per-op autograd bookkeeping on the host).
All activations are NHWC fp32.  Every arithmetic op is a libmtdgan_hip.so kernel (see kernels.py).
"""
import torch

from . import kernels as K
from .kernels import ACT_NONE, ACT_RELU

CH = 32


# ------------------------------------------------------------------------------------------------ block
def block_forward(x, w_img, b_img, w_fft, b_fft, save, w2t=None):
    """x: (B,64,64,32).  Returns (out, saved) with saved = (x, img, S, Z) when save."""
    B, H, W, _ = x.shape
    g = K.geom_fwd(B, H, W, 3, 1, 1)
    img = K.empty_nhwc(B, H, W, CH, x)
    if H != 64 or W != 64:
        # whole-slice inference (reference engine.py:89,129): LDS-resident transforms of side 128 / 256 / 512, forward only
        if save:
            raise RuntimeError("FFT_ConvBlock: training is implemented for 64 x 64 patches; larger maps are inference-only")
        if H != W or H not in (128, 256, 512):
            raise RuntimeError(f"FFT_ConvBlock: unsupported map size {H} x {W} (64, 128, 256 or 512 square)")
        out = K.empty_nhwc(B, H, W, CH, x)
        if K.conv_relu_add_ok(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, add1=x):
            # img = x + relu(conv3x3(x) + b) in one launch (MTD_ACT_RELU_ADD): the closing row transform then adds ONE operand
            K.conv(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, add1=x, act=K.ACT_RELU_ADD)
            K.spectral_branch_any(x, w2t if w2t is not None else K.transpose64(w_fft), b_fft, out, add1=img)
        else:
            K.conv(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, act=ACT_RELU)
            K.spectral_branch_any(x, w2t if w2t is not None else K.transpose64(w_fft), b_fft, out, add1=x, add2=img)
        return out, None
    if w2t is None:
        w2t = K.transpose64(w_fft)
    if K.BLOCK_FWD_WINO:
        # lab (round 6): the spatial branch on the persistent F(2x4, 3x3) kernel (conv_wino_c32.h, side stream) + the closing row
        # transform as a launch of its own, instead of the halo-tile kernel with the transform in its tail
        if K.BLOCK_FWD_WINO == 2:      # ... all on one stream (a persistent one-workgroup-per-CU kernel beside another stream's work waits for CUs)
            K.conv(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, act=ACT_RELU, wino32=True)
            R = K.rfft_rows(x, 0)
            T, S, Z = K.spec_mix_fwd(R, w2t, b_fft, save)
            out = K.empty_nhwc(B, H, W, CH, x)
        else:
            side = K.side_stream(x.device, 1)
            side.run(lambda: K.conv(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, act=ACT_RELU, wino32=True), x)
            R = K.rfft_rows(x, 0)
            T, S, Z = K.spec_mix_fwd(R, w2t, b_fft, save)
            out = K.empty_nhwc(B, H, W, CH, x)
            side.join()
        K.irfft_rows(T, out, add1=x, add2=img)
        return out, ((x, img, S, Z) if save else None)
    if K.BLOCK_TAIL and K.block_tail_ok(x, w_img, g, img, b_img):
        # spectral branch first; the spatial branch's launch then carries the inverse row transform and the residual:
        # img = relu(conv3x3(x) + b), out = x + img + irfft_rows(T) (mtd_resfft_block_tail)
        R = K.rfft_rows(x, 0)
        T, S, Z = K.spec_mix_fwd(R, w2t, b_fft, save)
        out = K.empty_nhwc(B, H, W, CH, x)
        K.block_tail(x, w_img, g, T, img, out, bias=b_img, act=ACT_RELU)
        return out, ((x, img, S, Z) if save else None)
    # the spatial branch (fp32-MFMA conv) runs on a side stream beside the spectral branch (FFT rows / columns)
    side = K.side_stream(x.device, 1)
    side.run(lambda: K.conv(x, w_img, g, CH, CH, CH * 9, 9, img, bias=b_img, act=ACT_RELU), x)   # relu(conv3x3(x)+b)
    R = K.rfft_rows(x, 0)
    T, S, Z = K.spec_mix_fwd(R, w2t, b_fft, save)
    out = K.empty_nhwc(B, H, W, CH, x)
    side.join()
    K.irfft_rows(T, out, add1=x, add2=img)                                          # x + img + irfft2(...)
    return out, ((x, img, S, Z) if save else None)


def block_backward(g, saved, w_img, w_fft, grads, premask, defer=None, gm=None):
    """g: grad of the block output.  grads: dict with tensors dw_img, db_img, dw_fft, db_fft (written).
    premask: multiply the input gradient by (x > 0) -- x is always a ReLU output inside the generator,
    so the result is the gradient w.r.t. the producer's pre-activation.  gm: g * (img > 0) if the launch that
    produced g has written it already (conv(out2=...))."""
    x, img, S, Z = saved
    B, H, W, _ = x.shape
    side = K.side_stream(x.device)
    if gm is None:
        gm = K.act_grad(g, img, 0.0)                                                # g * (img > 0)
    if K.BLOCK_TAIL and defer is not None and K.DEFER_WGRADS and not K.FUSE_WGRAD_ROWS and not K.BLOCK_BWD_WINO:
        # spectral chain first; the block conv's fused data + weight gradient launch then also takes the closing row
        # transform: gx = (dgrad(gm) + g + irfft_rows(gT)) * (x > 0)  (mtd_conv_c32_bwd_irfft)
        gx = K.empty_nhwc(B, H, W, CH, x)
        dg = ((gm, w_img, K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, gx),
              dict(add1=g, mask=x if premask else None, mask_slope=0.0))
        wg = ((gm, x, K.geom_fwd(B, H, W, 3, 1, 1), CH, CH, grads["dw_img"], CH * 9, 9), dict(db=grads["db_img"]))
        if K.conv_wgrad_fusable(dg, wg):
            gR = K.rfft_rows(g, 1)                                                  # irfft2 backward
            gT = K.spec_mix_bwd(gR, w_fft, S, Z, grads["dw_fft"], grads["db_fft"], defer=defer)
            if not K.conv_wgrad_fused(dg, wg, defer, spec=gT):      # (a launch: never inside an assert, python -O strips those)
                raise RuntimeError("block_backward: mtd_conv_c32_bwd_irfft refused a layer that mtd_conv_c32_bwd_ok accepted")
            return gx
    if K.BLOCK_BWD_WINO >= 2 and defer is not None and K.DEFER_WGRADS:
        # lab: the three-launch form with the spectral chain and the data gradient on ONE stream, the weight gradient beside them
        gR = K.rfft_rows(g, 1)
        gT = K.spec_mix_bwd(gR, w_fft, S, Z, grads["dw_fft"], grads["db_fft"], defer=defer)
        if K.BLOCK_BWD_WINO == 3:      # (3: the weight gradient on the same stream too)
            K.wgrad(gm, x, K.geom_fwd(B, H, W, 3, 1, 1), CH, CH, grads["dw_img"], CH * 9, 9, db=grads["db_img"], defer=defer)
        else:
            side.run(lambda: K.wgrad(gm, x, K.geom_fwd(B, H, W, 3, 1, 1), CH, CH, grads["dw_img"], CH * 9, 9, db=grads["db_img"], defer=defer), gm, g)
        d1 = K.empty_nhwc(B, H, W, CH, x)
        K.conv(gm, w_img, K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, d1, add1=g, wino32=True)
        gx = K.empty_nhwc(B, H, W, CH, x)
        K.irfft_rows(gT, gx, add1=d1, mask=x if premask else None)
        return gx
    # the block conv's weight gradient and the row transform that opens the spectral backward chain read the same
    # cotangent: one launch when the slab sums are deferred (kernels.wgrad rows=...), at the head of the spectral stream
    fused_rows = defer is not None and K.DEFER_WGRADS and K.FUSE_WGRAD_ROWS
    # data gradient + weight gradient of the block's 3x3 conv in ONE launch (csrc/conv_c32_bwd.hip) where the pair is eligible
    d1 = K.empty_nhwc(B, H, W, CH, x)
    dgrad_call = ((gm, w_img, K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, d1), dict(add1=g, wino32=True) if K.BLOCK_BWD_WINO else dict(add1=g))
    wgrad_args = (gm, x, K.geom_fwd(B, H, W, 3, 1, 1), CH, CH, grads["dw_img"], CH * 9, 9)
    fused_bwd = (not fused_rows) and K.conv_wgrad_fused(dgrad_call, (wgrad_args, dict(db=grads["db_img"])), defer)
    if not fused_rows and not fused_bwd:
        side.run(lambda: K.wgrad(*wgrad_args, db=grads["db_img"], defer=defer), gm, g)
    # spectral branch backward on a second side stream, beside the spatial data gradient on the main stream
    side1 = K.side_stream(x.device, 1)
    box = []

    def spectral():
        if fused_rows:
            gR = K.wgrad(gm, x, K.geom_fwd(B, H, W, 3, 1, 1), CH, CH, grads["dw_img"], CH * 9, 9, db=grads["db_img"], defer=defer,
                         rows=(g, 1))
        else:
            gR = K.rfft_rows(g, 1)                                                  # irfft2 backward
        box.append(K.spec_mix_bwd(gR, w_fft, S, Z, grads["dw_fft"], grads["db_fft"], defer=defer))
    side1.run(spectral, g, gm)
    if not fused_bwd:
        K.conv(*dgrad_call[0], **dgrad_call[1])                                     # dgrad(img branch) + residual
    gx = K.empty_nhwc(B, H, W, CH, x)
    side1.join()
    gT = box[0]
    K.crosses_streams(gT)
    K.irfft_rows(gT, gx, add1=d1, mask=x if premask else None)
    return gx


# ------------------------------------------------------------------------------------------------ generator
class GenParams:
    """Views of the generator parameters in the order the schedules use them."""

    def __init__(self, enc_w, enc_b, dec_w, dec_b, blk):
        self.enc_w, self.enc_b, self.dec_w, self.dec_b, self.blk = enc_w, enc_b, dec_w, dec_b, blk   # blk[i] = (w_img,b_img,w_fft,b_fft)


def generator_forward(x, P, save, out=None):
    """x: (B,64,64,1) NHWC.  Returns (out (B,64,64,1), tape).  out: optional destination of the result (train_step.d_loss
    hands in the second half of the discriminator's paired input batch: no concatenation pass)."""
    B, H, W, _ = x.shape
    L = len(P.enc_w) - 1                                   # 10
    views = []
    for i in range(1, L + 1):
        views.append((P.enc_w[i], CH, CH, CH * 9, 9))
        views.append((P.dec_w[i], CH, CH, 9, CH * 9))
        if save:
            views.append((P.enc_w[i], CH, CH, 9, CH * 9))
            views.append((P.dec_w[i], CH, CH, CH * 9, 9))
    for (w_img, _b, _w2, _b2) in P.blk:
        views.append((w_img, CH, CH, CH * 9, 9))
        if save:
            views.append((w_img, CH, CH, 9, CH * 9))
    K.prepack(views)                                       # one launch for all [tap][n][c] weight views
    w2ts = K.transpose64_all([blk[2] for blk in P.blk])      # ... and one for the mix weights (cached until they change)
    gf = K.geom_fwd(B, H, W, 3, 1, 1)
    gt = K.geom_dgrad_s1(B, H, W, 3, 1)                    # ConvTranspose2d(k3,s1,p1) gathers like a stride-1 dgrad
    # The plain encoder / decoder layers run on the persistent F(2x4, 3x3) kernel of the 32-channel layers where conv() takes
    # them (kernels.winograd_takes: whole slices always, the training patches in this forward pass): their transformed weights
    # in one launch per weight update.
    fwd32 = dict(wino32=True)
    wviews = ([(P.enc_w[i], CH, CH, CH * 9, 9, gf, fwd32) for i in range(1, L + 1)]
              + [(P.dec_w[i], CH, CH, 9, CH * 9, gt, fwd32) for i in range(1, L + 1)]
              + [(blk[0], CH, CH, CH * 9, 9, gf, fwd32 if K.BLOCK_FWD_WINO else {}) for blk in P.blk])      # (the blocks' convs: whole slices; training patches under BLOCK_FWD_WINO)
    if save:
        # ... and the views the backward pass will ask for (generator_backward / block_backward), in the same launch
        m32 = dict(wino32=True, mask=x)                    # (the keywords decide the routing, not the values)
        if K.WINO_C32_BWD:
            wviews += [(P.dec_w[i], CH, CH, CH * 9, 9, gf, m32) for i in range(1, L + 1)]
            wviews += [(P.enc_w[i], CH, CH, 9, CH * 9, gt, m32) for i in range(1, L + 1)]
        if K.BLOCK_BWD_WINO:
            wviews += [(blk[0], CH, CH, 9, CH * 9, gt, fwd32) for blk in P.blk]
    K.prepack_winograd(wviews)
    tape = {"t": [], "e": [], "blk": [], "d": [], "u": []}
    t = K.empty_nhwc(B, H, W, CH, x)
    K.conv(x, P.enc_w[0], gf, CH, 1, 9, 9, t, bias=P.enc_b[0], act=ACT_RELU)
    e = None
    for i in range(L + 1):
        e, sv = block_forward(t, *P.blk[i], save, w2t=w2ts.get(id(P.blk[i][2])))
        if save:
            tape["t"].append(t)
            tape["blk"].append(sv)
            tape["e"].append(e)
        else:
            tape["e"].append(e)
        if i < L:
            t = K.empty_nhwc(B, H, W, CH, x)
            K.conv(e, P.enc_w[i + 1], gf, CH, CH, CH * 9, 9, t, bias=P.enc_b[i + 1], act=ACT_RELU, wino32=True)
    # e list: e1..e10, xb  (index 0..10)
    cur = e                                                # x_b
    for j in range(L, 0, -1):                              # decoder[j], j = 10..1
        d = K.empty_nhwc(B, H, W, CH, x)
        K.conv(cur, P.dec_w[j], gt, CH, CH, 9, CH * 9, d, bias=P.dec_b[j], add1=tape["e"][j - 1], act=ACT_RELU, wino32=True)
        u, sv = block_forward(d, *P.blk[2 * L + 1 - j], save, w2t=w2ts.get(id(P.blk[2 * L + 1 - j][2])))   # enforce[11] after decoder[-1] ... enforce[20] after decoder[-10]
        if save:
            tape["u"].append(cur)                          # input of decoder[j]
            tape["d"].append(d)
            tape["blk"].append(sv)
        cur = u
    if out is None:
        out = K.empty_nhwc(B, H, W, 1, x)
    K.conv(cur, P.dec_w[0], gt, 1, CH, 9, 9, out, bias=P.dec_b[0], add1=x, act=ACT_RELU)
    if save:
        tape["u"].append(cur)
        tape["x"] = x
        tape["out"] = out
    return out, (tape if save else None)


def generator_backward(g_out, tape, P, G):
    """g_out: grad of the output (B,64,64,1).  G: GenParams-shaped container of gradient tensors
    (enc_w[i] ... blk[i] = dict(dw_img, db_img, dw_fft, db_fft)); all are overwritten."""
    x = tape["x"]
    B, H, W, _ = x.shape
    L = len(P.enc_w) - 1
    gf = K.geom_fwd(B, H, W, 3, 1, 1)
    gt = K.geom_dgrad_s1(B, H, W, 3, 1)
    # output ReLU
    side = K.side_stream(x.device)
    defer = K.DeferredWgrads()      # the 62 slab sums of this pass (41 conv, 21 mix layers) are taken in two launches at the end
    gpre = K.act_grad(g_out, tape["out"], 0.0)
    # decoder[0]: ConvTranspose 32 -> 1
    u0 = tape["u"][L]
    side.run(lambda: K.wgrad(gpre, u0, gt, 1, CH, G.dec_w[0], 9, 9, db=G.dec_b[0]), gpre)
    gu = K.empty_nhwc(B, H, W, CH, x)
    K.conv(gpre, P.dec_w[0], gf, CH, 1, 9, 9, gu)          # d/du0: plain conv with W_t read as OIHW [32][1][3][3]
    skip = [None] * (L + 1)
    # A data-gradient launch also writes its result times (img > 0) of the block that consumes it -- that block's
    # masked cotangent, otherwise a pass of its own (21 x 10 us per step) -- where the halo-tile kernel runs it.
    fuse = K.fuses_masked_cotangent(B, H, W, CH, CH)
    w32 = dict(wino32=True) if K.WINO_C32_BWD else {}      # (the plain layers' data gradients on the persistent F(2x4) kernel's MASKED2 form)
    gm = None
    # blocks 20..11 and decoders 1..10 (tape order: d/u/blk appended for j = 10..1)
    for j in range(1, L + 1):
        k = L - j                                           # position in the tape lists for decoder[j]
        gpre_d = block_backward(gu, tape["blk"][L + 1 + k], *_blk_w(P, 2 * L + 1 - j), G.blk[2 * L + 1 - j], True, defer, gm)
        skip[j] = gpre_d                                    # flows unchanged into e_j
        uj = tape["u"][k]                                   # input of decoder[j]
        gu = K.empty_nhwc(B, H, W, CH, x)
        wg = ((gpre_d, uj, gt, CH, CH, G.dec_w[j], 9, CH * 9), dict(db=G.dec_b[j]))
        if fuse:                                            # consumer: block 2L - j (tape position L + k), or block L after the loop
            gm = K.empty_nhwc(B, H, W, CH, x)
            dg = ((gpre_d, P.dec_w[j], gf, CH, CH, CH * 9, 9, gm), dict(mask=tape["blk"][L + k][1], mask_slope=0.0, out2=gu, **w32))
        else:
            dg = ((gpre_d, P.dec_w[j], gf, CH, CH, CH * 9, 9, gu), dict(w32))
        if not K.conv_wgrad_fused(dg, wg, defer):           # one launch for the layer's two gradients, else two
            side.run(lambda: K.wgrad(*wg[0], db=G.dec_b[j], defer=defer), gpre_d)
            K.conv(*dg[0], **dg[1])
    # gu is now the gradient of x_b (output of block 10)
    g_e = gu
    for i in range(L, -1, -1):                              # blocks 10..0, encoders 10..0
        gpre_t = block_backward(g_e, tape["blk"][i], *_blk_w(P, i), G.blk[i], True, defer, gm)
        if i > 0:
            e_prev = tape["e"][i - 1]
            g_e = K.empty_nhwc(B, H, W, CH, x)
            wg = ((gpre_t, e_prev, gf, CH, CH, G.enc_w[i], CH * 9, 9), dict(db=G.enc_b[i]))
            if fuse:
                gm = K.empty_nhwc(B, H, W, CH, x)
                dg = ((gpre_t, P.enc_w[i], K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, gm),
                      dict(add1=skip[i], mask=tape["blk"][i - 1][1], mask_slope=0.0, out2=g_e, **w32))
            else:
                dg = ((gpre_t, P.enc_w[i], K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, g_e), dict(add1=skip[i], **w32))
            if not K.conv_wgrad_fused(dg, wg, defer):
                side.run(lambda: K.wgrad(*wg[0], db=G.enc_b[i], defer=defer), gpre_t)
                K.conv(*dg[0], **dg[1])
        else:
            side.run(lambda: K.wgrad(gpre_t, x, gf, CH, 1, G.enc_w[0], 9, 9, db=G.enc_b[0]), gpre_t)
    side.join()
    K.side_stream(x.device, 1).join()
    K.flush_wgrads(defer)


def _blk_w(P, i):
    w_img, b_img, w_fft, b_fft = P.blk[i]
    return w_img, w_fft


# ------------------------------------------------------------------------------------------------ RED-CNN generator
def redcnn_forward(x, enc_w, enc_b, dec_w, dec_b, save):
    """reference arch/Ours/networks.py:498-505 (REDCNN_Generator.forward, the ablation family's generator):
        residuals = inputs of the 11 encoder convs; x = relu(enc_k(x)); then, decoders in reverse,
        x = relu(dec_k(x) + residual_k).
    x: (B,64,64,1) NHWC.  Returns (out (B,64,64,1), tape).  Layers 1..10 are the 32 -> 32 channel 3x3 kernels of the
    Res-FFT generator (halo-tile implicit GEMM), layer 0 the 1 -> 32 / 32 -> 1 vector kernels."""
    B, H, W, _ = x.shape
    L = len(enc_w) - 1
    views = []
    for i in range(1, L + 1):
        views.append((enc_w[i], CH, CH, CH * 9, 9))
        views.append((dec_w[i], CH, CH, 9, CH * 9))
        if save:
            views.append((enc_w[i], CH, CH, 9, CH * 9))
            views.append((dec_w[i], CH, CH, CH * 9, 9))
    K.prepack(views)
    gf = K.geom_fwd(B, H, W, 3, 1, 1)
    gt = K.geom_dgrad_s1(B, H, W, 3, 1)
    t = [x]                                                 # t[k]: input of encoder k (the residual of decoder k)
    for k in range(L + 1):
        o = K.empty_nhwc(B, H, W, CH, x)
        K.conv(t[k], enc_w[k], gf, CH, 1 if k == 0 else CH, (1 if k == 0 else CH) * 9, 9, o, bias=enc_b[k], act=ACT_RELU)
        t.append(o)
    u = {L + 1: t[L + 1]}                                   # u[k]: input of decoder k - 1 ... u[L + 1] = last encoder output
    cur = t[L + 1]
    for k in range(L, -1, -1):
        n_out = 1 if k == 0 else CH
        o = K.empty_nhwc(B, H, W, n_out, x)
        K.conv(cur, dec_w[k], gt, n_out, CH, 9, n_out * 9, o, bias=dec_b[k], add1=t[k], act=ACT_RELU)
        u[k] = o
        cur = o
    return cur, ({"t": t, "u": u, "x": x} if save else None)


def redcnn_backward(g_out, tape, enc_w, dec_w, G_enc_w, G_enc_b, G_dec_w, G_dec_b):
    """Autograd transpose of redcnn_forward: every parameter gradient is overwritten.  The 32 -> 32 layers use the fused
    data + weight gradient launch (kernels.conv_wgrad_fused) where eligible."""
    t, u, x = tape["t"], tape["u"], tape["x"]
    B, H, W, _ = x.shape
    L = len(enc_w) - 1
    gf = K.geom_fwd(B, H, W, 3, 1, 1)
    gt = K.geom_dgrad_s1(B, H, W, 3, 1)
    side = K.side_stream(x.device)
    defer = K.DeferredWgrads()
    gpre = K.act_grad(g_out, u[0], 0.0)                     # through the output ReLU
    # decoder 0: ConvTranspose 32 -> 1 (its residual is the network input: no gradient needed)
    u1 = u[1]
    side.run(lambda: K.wgrad(gpre, u1, gt, 1, CH, G_dec_w[0], 9, 9, db=G_dec_b[0]), gpre)
    g = K.empty_nhwc(B, H, W, CH, x)
    K.conv(gpre, dec_w[0], gf, CH, 1, 9, 9, g, mask=u[1], mask_slope=0.0)      # gradient of u[1]'s pre-activation
    skip = [None] * (L + 2)
    for k in range(1, L + 1):                               # decoders 1..L: pre-activation gradient g of u[k]
        skip[k] = g                                         # flows unchanged into the residual t[k]
        nxt = K.empty_nhwc(B, H, W, CH, x)
        src = u[k + 1]                                      # input of decoder k (u[L + 1] = t[L + 1])
        wg = ((g, src, gt, CH, CH, G_dec_w[k], 9, CH * 9), dict(db=G_dec_b[k]))
        kw = dict(mask=src, mask_slope=0.0)                 # times (input > 0): the pre-activation gradient of the producer
        dg = ((g, dec_w[k], gf, CH, CH, CH * 9, 9, nxt), kw)
        if not K.conv_wgrad_fused(dg, wg, defer):
            side.run(lambda wg=wg, k=k: K.wgrad(*wg[0], db=G_dec_b[k], defer=defer), g)
            K.conv(*dg[0], **dg[1])
        g = nxt
    # g: pre-activation gradient of t[L + 1] (the last encoder's output feeds decoder L only)
    for k in range(L, -1, -1):                              # encoders L..0; encoder k maps t[k] -> t[k + 1]
        if k > 0:
            nxt = K.empty_nhwc(B, H, W, CH, x)
            wg = ((g, t[k], gf, CH, CH, G_enc_w[k], CH * 9, 9), dict(db=G_enc_b[k]))
            dg = ((g, enc_w[k], K.geom_dgrad_s1(B, H, W, 3, 1), CH, CH, 9, CH * 9, nxt), dict(add1=skip[k], mask=t[k], mask_slope=0.0))
            if not K.conv_wgrad_fused(dg, wg, defer):
                side.run(lambda wg=wg, k=k: K.wgrad(*wg[0], db=G_enc_b[k], defer=defer), g)
                K.conv(*dg[0], **dg[1])
            g = nxt
        else:
            side.run(lambda g=g: K.wgrad(g, x, gf, CH, 1, G_enc_w[0], 9, 9, db=G_enc_b[0]), g)
    side.join()
    K.flush_wgrads(defer)

"""Forward / backward kernel schedules of Multi_Task_Discriminator_Skip on the HIP kernels.

Mirrors the reference's arch/Ours/networks.py:383-474 (forward) and spells out its autograd transpose
so that a whole discriminator pass is one node: a pass is recorded on a tape, and `disc_backward` replays
it for any combination of output cotangents (image-level score, pixel-level map, restoration), writing
parameter gradients straight into caller-owned buffers (this is what lets PCGrad keep one flat gradient
vector per task).  Spectral norm: sigma per pass comes from mtd_sn_power_iter; convolutions read
weight_orig in place and scale their accumulators by 1/sigma; the weight gradient correction
 is one batched mtd_sn_grad per backward.  NHWC fp32 throughout.
"""
from . import _options
import ctypes as C
import os

import torch

from . import _lib
from . import kernels as K
from .kernels import ACT_LRELU, ACT_NONE

PS_FUSED = _options.lab("MTD_NO_PS_FUSE", "0") != "1"     # r_up{l}: conv1x1 + PixelShuffle as four strided-output classes in one grid
CH = [64, 128, 256, 512, 512, 512]                    # trunk channels per level (out_channels = 64)
DEC = [(1024, 512), (1024, 512), (1024, 256), (512, 128), (256, 64), (128, 1)]   # (cat channels, out) per decoder level
RUP = [(512, 512), (512, 512), (512, 512), (256, 256), (128, 128), (64, 64)]      # r_up{l}: cin -> cout' (conv to 4*cout')


def sn_layer_specs():
    """(name, rows, cols) of the 45 spectral-normalised layers in a fixed order."""
    out = []
    cin = 1
    for l, co in enumerate(CH, start=1):
        out.append((f"conv{l}1", co, cin * 9))
        out.append((f"conv{l}2", co, co * 9))
        out.append((f"down{l}", co, co * 16))
        cin = co
    out += [("bconv1", 512, 512), ("bconv2", 512, 512), ("c_fc", 512, 512)]
    for pre in ("s", "r"):
        for l, (ci, co) in enumerate(DEC, start=1):
            out.append((f"{pre}_dconv{l}1", co, ci * 9))
            out.append((f"{pre}_dconv{l}2", co, co * 9))
    return out


SN_SPECS = sn_layer_specs()
SN_INDEX = {n: i for i, (n, _, _) in enumerate(SN_SPECS)}
SN_ROW_OFF, SN_COL_OFF = [], []
_r = _c = 0
for _n, _rows, _cols in SN_SPECS:
    SN_ROW_OFF.append(_r)
    SN_COL_OFF.append(_c)
    _r += _rows
    _c += _cols
SN_ROWS_TOTAL, SN_COLS_TOTAL = _r, _c
SN_W_OFF = []
_o = 0
for _n, _rows, _cols in SN_SPECS:
    SN_W_OFF.append(_o)
    _o += _rows * _cols
SN_W_TOTAL = _o


def conv_views(P, backward):
    """Every implicit-GEMM weight view a discriminator pass (and its replay) will ask kernels.conv() for, so
    that they can be packed in one launch per optimizer step instead of one per layer."""
    v = []

    def add(name, N, Cc, k, suffix=".weight_orig", r=0):
        w = P.get(name + suffix)
        if w is None:                                          # a head this discriminator does not have (ablation subsets)
            return
        # (3x3 layers on r x r maps that kernels.conv() sends to the Winograd kernel need no [tap][n][c] view: winograd_views)
        if not (r and K.winograd_takes(K.geom_fwd(1, r, r, 3, 1, 1), N, Cc, {})):
            v.append((w, N, Cc, Cc * k * k, k * k))           # forward view (OIHW)
        if backward and not (r and K.winograd_takes(K.geom_dgrad_s1(1, r, r, 3, 1), Cc, N, {})):
            v.append((w, Cc, N, k * k, Cc * k * k))           # data-gradient view (transposed)

    cin, r = 1, 64
    for l, co in enumerate(CH, start=1):
        add(f"conv{l}1", co, cin, 3, r=r)
        add(f"conv{l}2", co, co, 3, r=r)
        add(f"down{l}", co, co, 4)
        cin, r = co, r // 2
    add("bconv1", 512, 512, 1)
    add("bconv2", 512, 512, 1)
    add("c_fc", 512, 512, 1)
    for pre in ("s", "r"):
        r = 2
        for l, (ci, co) in enumerate(DEC, start=1):
            add(f"{pre}_dconv{l}1", co, ci, 3, r=r)
            add(f"{pre}_dconv{l}2", co, co, 3, r=r)
            r *= 2
    for l, (ci, cu) in enumerate(RUP, start=1):
        add(f"r_up{l}.upsample.0", 4 * cu, ci, 1, ".weight")
    return v


def winograd_views(P, backward):
    """The 3x3 stride-1 layers whose maps are large enough for kernels.conv() to send them to the Winograd kernel
    (kernels.winograd_takes), as (w, N, C, w_sn, w_sc, geometry) for kernels.prepack_winograd: their transformed weights are
    then built in ONE launch per optimizer step.  The geometry only selects the tap order (forward / data gradient) and the
    size thresholds, so batch 1 stands in for the real batch."""
    v = []

    def add(name, N, Cc, r):
        w = P.get(name + ".weight_orig")
        if w is None:
            return
        v.append((w, N, Cc, Cc * 9, 9, K.geom_fwd(1, r, r, 3, 1, 1)))
        if backward:
            v.append((w, Cc, N, 9, Cc * 9, K.geom_dgrad_s1(1, r, r, 3, 1)))

    cin, r = 1, 64
    for l, co in enumerate(CH, start=1):
        add(f"conv{l}1", co, cin, r)
        add(f"conv{l}2", co, co, r)
        cin, r = co, r // 2
    for pre in ("s", "r"):
        r = 2
        for l, (ci, co) in enumerate(DEC, start=1):
            add(f"{pre}_dconv{l}1", co, ci, r)
            add(f"{pre}_dconv{l}2", co, co, r)
            r *= 2
    return v


def winograd_s2_views(P, backward):
    """The 4x4 / stride-2 `down` layers in the views kernels.conv() (forward) and kernels.conv_multi() (the four parity classes of
    the data gradient) ask the F(3x3, 2x2) kernel's weights in, for kernels.prepack_winograd_s2 (which prepares the ones the
    step really uses).  The geometry only selects the filter entries, so batch 1 stands in for the real batch."""
    v = []
    r = 64
    for l, co in enumerate(CH, start=1):
        w = P.get(f"down{l}.weight_orig")
        if w is not None and r // 2 >= K.WINO_S2_MIN_HW:
            v.append((w, co, co, co * 16, 16, K.geom_fwd(1, r, r, 4, 2, 1)))
            if backward:
                for py in range(2):
                    for px in range(2):
                        v.append((w, co, co, 16, co * 16, K.geom_dgrad_s2(1, r, r, py, px)))
        r //= 2
    return v


class DiscRuntime:
    """Per-module scratch: the raw weight-gradient temp (one flat buffer shared by all SN layers)."""

    def __init__(self):
        self._gtemp = {}

    def gtemp(self, name, device, which=0):
        """which: 0 / 1 = first / second pass of a paired tape (each pass has its own sigma, u, v)."""
        buf = self._gtemp.get(which)
        if buf is None or buf.device != device:
            buf = torch.empty(SN_W_TOTAL, dtype=torch.float32, device=device)
            self._gtemp[which] = buf
        i = SN_INDEX[name]
        return buf[SN_W_OFF[i]:SN_W_OFF[i] + SN_SPECS[i][1] * SN_SPECS[i][2]]


class Tape:
    pass


SN_MERGE_HALVES = _options.lab("MTD_SN_MERGE_HALVES", "1") != "0"      # ... and both halves of a paired pass in one small-map weight-gradient launch
SN_ACT_DOT = _options.lab("MTD_SN_ACT_DOT", "1") != "0"        # the spectral-norm correction's <G, W> from (cotangent, saved output) on the small maps (round 6)
SKIP_IN_CAT = _options.lab("MTD_SKIP_IN_CAT", "1") != "0"      # trunk outputs written into the pixel-level decoder's concatenated buffers


def _sn_forward(P, train, device):
    """Power iteration + sigma for all 45 layers (4 launches).  Returns (sig[45,2], u_save, v_save)."""
    L = _lib.lib()
    sig = torch.empty((len(SN_SPECS), 2), dtype=torch.float32, device=device)
    u_save = torch.empty(SN_ROWS_TOTAL, dtype=torch.float32, device=device)
    v_save = torch.empty(SN_COLS_TOTAL, dtype=torch.float32, device=device)
    structs = []
    for i, (n, rows, cols) in enumerate(SN_SPECS):
        if n + ".weight_orig" not in P:                        # ablation subsets: no such layer, its sigma slot stays unused
            continue
        s = _lib.SnLayer()
        s.w = P[n + ".weight_orig"].data_ptr()
        s.u = P[n + ".weight_u"].data_ptr()
        s.v = P[n + ".weight_v"].data_ptr()
        s.sigma = sig.data_ptr() + 8 * i
        s.u_save = u_save.data_ptr() + 4 * SN_ROW_OFF[i]
        s.v_save = v_save.data_ptr() + 4 * SN_COL_OFF[i]
        s.rows, s.cols = rows, cols
        structs.append(s)
    dev_tab, host_arr = K.device_table(structs, device)
    need = L.mtd_sn_ws_bytes(C.cast(host_arr, C.c_void_p), len(structs))
    ws = K.workspace(need, device)
    K.check(L.mtd_sn_power_iter(dev_tab.data_ptr(), C.cast(host_arr, C.c_void_p), len(structs), 1 if train else 0, ws.data_ptr(),
                                K.stream_ptr()), "mtd_sn_power_iter")
    return sig, u_save, v_save


SN_FUSED_ITERS = _options.lab("MTD_SN_FUSED_ITERS", "1") != "0"      # d_loss: its four power iterations share passes over the weights (round 6)


def _sn_forward_multi(P, train, device, nit):
    """`nit` consecutive power iterations on the same weights -- the discriminator step's four passes, which the reference runs before
    any weight changes (networks.py:1957-1992).  Returns [(sig, u_save, v_save)] * nit like nit calls of _sn_forward; in train mode
    they go through mtd_sn_power_iter_multi: W v of one iteration and W^T (W v) of the next share one pass over the 260 MB of
    weights (nit + 1 passes instead of 2 nit)."""
    if not (train and SN_FUSED_ITERS and nit > 1):
        return [_sn_forward(P, train, device) for _ in range(nit)]
    L = _lib.lib()
    outs, structs = [], []
    for _it in range(nit):
        sig = torch.empty((len(SN_SPECS), 2), dtype=torch.float32, device=device)
        u_save = torch.empty(SN_ROWS_TOTAL, dtype=torch.float32, device=device)
        v_save = torch.empty(SN_COLS_TOTAL, dtype=torch.float32, device=device)
        outs.append((sig, u_save, v_save))
        for i, (n, rows, cols) in enumerate(SN_SPECS):
            if n + ".weight_orig" not in P:
                continue
            s = _lib.SnLayer()
            s.w = P[n + ".weight_orig"].data_ptr()
            s.u = P[n + ".weight_u"].data_ptr()
            s.v = P[n + ".weight_v"].data_ptr()
            s.sigma = sig.data_ptr() + 8 * i
            s.u_save = u_save.data_ptr() + 4 * SN_ROW_OFF[i]
            s.v_save = v_save.data_ptr() + 4 * SN_COL_OFF[i]
            s.rows, s.cols = rows, cols
            structs.append(s)
    n_layers = len(structs) // nit
    dev_tab, host_arr = K.device_table(structs, device)
    need = L.mtd_sn_ws_bytes(C.cast(host_arr, C.c_void_p), n_layers)
    ws = K.workspace(need, device)
    K.check(L.mtd_sn_power_iter_multi(dev_tab.data_ptr(), C.cast(host_arr, C.c_void_p), n_layers, nit, ws.data_ptr(), K.stream_ptr()),
            "mtd_sn_power_iter_multi")
    return outs


def _inv_sigma(tape, name):
    i = SN_INDEX[name]
    return tape.sig[i, 1:2]


def _scales(tape, name, geom):
    """Accumulator scale(s) 1/sigma of an SN layer for a launch with geometry `geom`: a paired tape (two passes stacked along
    the batch, each with its own power-iteration state) switches to the second pass's 1/sigma at its first pixel."""
    i = SN_INDEX[name]
    kw = {"scale": tape.sig[i, 1:2]}
    if getattr(tape, "pair", 0):
        kw["scale2"] = tape.sig2[i, 1:2]
        kw["scale_split"] = tape.pair * geom.OH * geom.OW
    return kw


def _sn_call(P, tape, name, x, out, geom, N, Cc, k, act):
    """The (args, keywords) of an SN layer's forward conv as kernels.conv / conv_pair take them."""
    return ((x, P[name + ".weight_orig"], geom, N, Cc, Cc * k * k, k * k, out), dict(bias=P[name + ".bias"], act=act, **_scales(tape, name, geom)))


def _sn_conv(P, tape, name, x, out, geom, N, Cc, k, act):
    args, kw = _sn_call(P, tape, name, x, out, geom, N, Cc, k, act)
    return K.conv(*args, **kw)


# Round 6: the pixel-level and the restoration decoder are mirrors (networks.py:420-467: s_dconv{l}k and r_dconv{l}k have the same
# channels on the same maps) -- where a pass runs both, level by level, each pair of mirror convs is ONE launch (kernels.conv_pair:
# two problems in one grid of the Winograd kernel, one slab-sum launch for both) in the forward pass and for the data gradients.
PAIR_DECODERS = _options.lab("MTD_PAIR_DECODERS", "1") != "0"


def disc_forward(P, x, train, drop_mask, need_rec, save, sn=None, sn2=None, pair=0, heads=("cls", "seg", "rec")):
    """x: (B,64,64,1) NHWC.  P: dict name -> tensor (reference state_dict names).  drop_mask: (B,512)
    multiplier or None.  Returns ((enc (B,1,1,1), dec (B,64,64,1), rec or None), tape).
    heads: which of the image-level head ("cls"), the bilinear pixel-level decoder ("seg") and the PixelShuffle
    restoration decoder ("rec") this discriminator has -- the reference's ablation discriminators (networks.py:507-1322:
    CLS_, SEG_, CLS_SEG_, CLS_REC_, SEG_REC_Discriminator) are the same trunk with a subset of them; outputs of absent
    heads are None.
    Paired mode (train_step.d_loss): x stacks TWO passes of the reference along the batch, the first `pair` images being
    the earlier pass; sn / sn2 are their power-iteration results (sigma table, u, v), run ahead of time in the reference's
    order.  Every conv then serves both passes in one launch with the 1/sigma of each half (scale2 / scale_split)."""
    B = x.shape[0]
    dev = x.device
    tp = Tape()
    K.prepack(conv_views(P, save))
    K.prepack_winograd(winograd_views(P, save))
    K.prepack_winograd_s2(winograd_s2_views(P, save))
    tp.sig, tp.u_save, tp.v_save = sn if sn is not None else _sn_forward(P, train, dev)
    tp.pair = int(pair)
    if tp.pair:
        if sn is None or sn2 is None or not (0 < tp.pair < B):
            raise ValueError("paired discriminator pass needs both power-iteration states and 0 < pair < batch")
        tp.sig2, tp.u_save2, tp.v_save2 = sn2
    tp.x_in, tp.B, tp.drop_mask, tp.need_rec = x, B, drop_mask, need_rec
    tp.tin, tp.a, tp.xs = {}, {}, {}
    t, h, cin = x, 64, 1
    # A trunk level's output is the skip operand of BOTH decoders' level 7 - l (networks.py:420-467: torch.cat([up, skip], 1)).  It is
    # written straight into the channel slice [ccat - co, ccat) of the pixel-level decoder's concatenated buffer (every kernel takes a
    # pixel stride): one copy of the skip per level and pass instead of two (18 launches of 3-25 us per iteration less).
    seg_cat = {}
    for l, co in enumerate(CH, start=1):
        g3 = K.geom_fwd(B, h, h, 3, 1, 1)
        a = K.empty_nhwc(B, h, h, co, x)
        _sn_conv(P, tp, f"conv{l}1", t, a, g3, co, cin, 3, ACT_LRELU)
        if SKIP_IN_CAT and "seg" in heads:
            ccat = DEC[6 - l][0]
            seg_cat[7 - l] = K.empty_nhwc(B, h, h, ccat, x)
            xl = seg_cat[7 - l][..., ccat - co:]
        else:
            xl = K.empty_nhwc(B, h, h, co, x)
        _sn_conv(P, tp, f"conv{l}2", a, xl, g3, co, co, 3, ACT_LRELU)
        d = K.empty_nhwc(B, h // 2, h // 2, co, x)
        _sn_conv(P, tp, f"down{l}", xl, d, K.geom_fwd(B, h, h, 4, 2, 1), co, co, 4, ACT_NONE)
        tp.tin[l], tp.a[l], tp.xs[l] = t, a, xl
        t, h, cin = d, h // 2, co
    g1 = K.geom_fwd(B, 1, 1, 1, 1, 0)
    tp.d6 = t
    tp.b1 = K.empty_nhwc(B, 1, 1, 512, x)
    _sn_conv(P, tp, "bconv1", t, tp.b1, g1, 512, 512, 1, ACT_LRELU)
    tp.bot = K.empty_nhwc(B, 1, 1, 512, x)
    _sn_conv(P, tp, "bconv2", tp.b1, tp.bot, g1, 512, 512, 1, ACT_LRELU)
    # ---- CLS head (networks.py:414-417, 470)
    enc = None
    if "cls" in heads:
        tp.c = K.empty_nhwc(B, 1, 1, 512, x)
        _sn_conv(P, tp, "c_fc", tp.bot, tp.c, g1, 512, 512, 1, ACT_LRELU)
        tp.cm = K.mul(tp.c, drop_mask.reshape(B, 1, 1, 512)) if drop_mask is not None else tp.c
        enc = K.empty_nhwc(B, 1, 1, 1, x)
        K.conv(tp.cm, P["enc_out.weight"], g1, 1, 512, 512, 1, enc, bias=P["enc_out.bias"])
    # ---- both decoders level by level, their mirror convs in pairs (PAIR_DECODERS)
    both = PAIR_DECODERS and "seg" in heads and need_rec and "rec" in heads
    g64 = K.geom_fwd(B, 64, 64, 1, 1, 0)
    if both:
        tp.s_cat, tp.s_o1, tp.s_o2, tp.s_in = {}, {}, {}, {}
        tp.r_cat, tp.r_o1, tp.r_o2, tp.r_in = {}, {}, {}, {}
        ts, tr, r = tp.bot, tp.bot, 1
        for lvl in range(1, 7):
            skip = tp.xs[7 - lvl]
            ccat, co = DEC[lvl - 1]
            cprev = ts.shape[3]
            cat_s = seg_cat.get(lvl)
            if cat_s is None:
                cat_s = K.empty_nhwc(B, 2 * r, 2 * r, ccat, x)
                K.copy_channels(skip, cat_s[..., cprev:])
            K.upsample2x_fwd(ts, cat_s[..., :cprev])
            cin_up, cup = RUP[lvl - 1]
            cat_r = K.empty_nhwc(B, 2 * r, 2 * r, ccat, x)
            wu = P[f"r_up{lvl}.upsample.0.weight"]
            if PS_FUSED:
                bu = K.regrouped_bias(P[f"r_up{lvl}.upsample.0.bias"], 4)
                wq = wu.detach().view(cup, 4, cin_up)
                K.conv_multi([((tr, wq[:, q], K.geom_pixel_shuffle2(B, r, r, q >> 1, q & 1), cup, cin_up, 4 * cin_up, 1, cat_r[..., :cup]),
                               dict(bias=bu[q * cup:(q + 1) * cup])) for q in range(4)])
            else:
                up = K.empty_nhwc(B, r, r, 4 * cup, x)
                K.conv(tr, wu, K.geom_fwd(B, r, r, 1, 1, 0), 4 * cup, cin_up, cin_up, 1, up, bias=P[f"r_up{lvl}.upsample.0.bias"])
                K.pixel_shuffle2_fwd(up, cat_r[..., :cup])
            K.copy_channels(skip, cat_r[..., cup:])
            r *= 2
            g3 = K.geom_fwd(B, r, r, 3, 1, 1)
            o1s, o1r = K.empty_nhwc(B, r, r, co, x), K.empty_nhwc(B, r, r, co, x)
            K.conv_pair(_sn_call(P, tp, f"s_dconv{lvl}1", cat_s, o1s, g3, co, ccat, 3, ACT_LRELU),
                        _sn_call(P, tp, f"r_dconv{lvl}1", cat_r, o1r, g3, co, ccat, 3, ACT_LRELU))
            o2s, o2r = K.empty_nhwc(B, r, r, co, x), K.empty_nhwc(B, r, r, co, x)
            K.conv_pair(_sn_call(P, tp, f"s_dconv{lvl}2", o1s, o2s, g3, co, co, 3, ACT_LRELU),
                        _sn_call(P, tp, f"r_dconv{lvl}2", o1r, o2r, g3, co, co, 3, ACT_LRELU))
            tp.s_in[lvl], tp.s_cat[lvl], tp.s_o1[lvl], tp.s_o2[lvl] = ts, cat_s, o1s, o2s
            tp.r_in[lvl], tp.r_cat[lvl], tp.r_o1[lvl], tp.r_o2[lvl] = tr, cat_r, o1r, o2r
            ts, tr = o2s, o2r
        dec = K.empty_nhwc(B, 64, 64, 1, x)
        K.conv(ts, P["dec_out.weight"], g64, 1, 1, 1, 1, dec, bias=P["dec_out.bias"])
        rec = K.empty_nhwc(B, 64, 64, 1, x)
        K.conv(tr, P["rec_out.weight"], g64, 1, 1, 1, 1, rec, bias=P["rec_out.bias"])
        return (enc, dec, rec), (tp if save else None)
    # ---- SEG decoder (networks.py:420-442)
    tp.s_cat, tp.s_o1, tp.s_o2, tp.s_in = {}, {}, {}, {}
    t, r = tp.bot, 1
    for lvl in range(1, 7) if "seg" in heads else ():
        r *= 2
        cprev = t.shape[3]
        skip = tp.xs[7 - lvl]
        ccat, co = DEC[lvl - 1]
        cat = seg_cat.get(lvl)
        if cat is None:
            cat = K.empty_nhwc(B, r, r, ccat, x)
            K.copy_channels(skip, cat[..., cprev:])
        K.upsample2x_fwd(t, cat[..., :cprev])
        g3 = K.geom_fwd(B, r, r, 3, 1, 1)
        o1 = K.empty_nhwc(B, r, r, co, x)
        _sn_conv(P, tp, f"s_dconv{lvl}1", cat, o1, g3, co, ccat, 3, ACT_LRELU)
        o2 = K.empty_nhwc(B, r, r, co, x)
        _sn_conv(P, tp, f"s_dconv{lvl}2", o1, o2, g3, co, co, 3, ACT_LRELU)
        tp.s_in[lvl], tp.s_cat[lvl], tp.s_o1[lvl], tp.s_o2[lvl] = t, cat, o1, o2
        t = o2
    dec = None
    if "seg" in heads:
        dec = K.empty_nhwc(B, 64, 64, 1, x)
        K.conv(t, P["dec_out.weight"], g64, 1, 1, 1, 1, dec, bias=P["dec_out.bias"])
    # ---- REC decoder (networks.py:445-467)
    rec = None
    if need_rec and "rec" in heads:
        tp.r_cat, tp.r_o1, tp.r_o2, tp.r_in = {}, {}, {}, {}
        t, r = tp.bot, 1
        for lvl in range(1, 7):
            cin_up, cup = RUP[lvl - 1]
            skip = tp.xs[7 - lvl]
            ccat, co = DEC[lvl - 1]
            cat = K.empty_nhwc(B, 2 * r, 2 * r, ccat, x)
            # conv1x1 (C -> 4C') + PixelShuffle(2) without the intermediate tensor: the four sub-pixel classes (i, j) are four
            # 1x1 convs over output channels c*4 + 2i + j (weight rows 4 apart) that write every other pixel of the
            # concatenated buffer -- one grid (conv_multi), like the parity classes of the stride-2 data gradients
            wu = P[f"r_up{lvl}.upsample.0.weight"]
            if PS_FUSED:
                bu = K.regrouped_bias(P[f"r_up{lvl}.upsample.0.bias"], 4)
                wq = wu.detach().view(cup, 4, cin_up)
                K.conv_multi([((t, wq[:, q], K.geom_pixel_shuffle2(B, r, r, q >> 1, q & 1), cup, cin_up, 4 * cin_up, 1, cat[..., :cup]),
                               dict(bias=bu[q * cup:(q + 1) * cup])) for q in range(4)])
            else:
                up = K.empty_nhwc(B, r, r, 4 * cup, x)
                K.conv(t, wu, K.geom_fwd(B, r, r, 1, 1, 0), 4 * cup, cin_up, cin_up, 1, up, bias=P[f"r_up{lvl}.upsample.0.bias"])
                K.pixel_shuffle2_fwd(up, cat[..., :cup])
            r *= 2
            K.copy_channels(skip, cat[..., cup:])
            g3 = K.geom_fwd(B, r, r, 3, 1, 1)
            o1 = K.empty_nhwc(B, r, r, co, x)
            _sn_conv(P, tp, f"r_dconv{lvl}1", cat, o1, g3, co, ccat, 3, ACT_LRELU)
            o2 = K.empty_nhwc(B, r, r, co, x)
            _sn_conv(P, tp, f"r_dconv{lvl}2", o1, o2, g3, co, co, 3, ACT_LRELU)
            tp.r_in[lvl], tp.r_cat[lvl], tp.r_o1[lvl], tp.r_o2[lvl] = t, cat, o1, o2
            t = o2
        rec = K.empty_nhwc(B, 64, 64, 1, x)
        K.conv(t, P["rec_out.weight"], g64, 1, 1, 1, 1, rec, bias=P["rec_out.bias"])
    return (enc, dec, rec), (tp if save else None)


class GradSink:
    """Where a backward pass accumulates parameter gradients: name -> pre-zeroed tensor (same shape as the
    parameter).  Names that are absent get no gradient (their weight-gradient kernels are skipped)."""

    def __init__(self, tensors):
        self.t = tensors

    def get(self, name):
        return self.t.get(name)


FLUSH_LEVEL = 4      # trunk levels 6 .. FLUSH_LEVEL + both bottleneck convs: 91 % of the shared parameters, done 60 % into the trunk
FLUSH_STAGES = tuple(_options.lab("MTD_DP_SHIP_STAGES", "heads,trunk_low").split(","))      # lab: which of the two early-shipping points are live


def disc_backward(*args, **kw):
    """One backward pass over a recorded tape: _disc_backward_gen run by itself (every launch it hands out is issued at once)."""
    gen = _disc_backward_gen(*args, **kw)
    try:
        call = next(gen)
        while True:
            K.conv(*call[0], **call[1])
            call = next(gen)
    except StopIteration as e:
        return e.value


# 0: one pass after the other; 1: the adversarial and the first consistency pass advanced together (the default); 3 (lab): the
# restoration pass with them -- measured on one box (three rounds): 26.21 / 25.90 / 26.16 ms: groups of three give most of the pairs' gain back
LOCKSTEP = int(_options.lab("MTD_LOCKSTEP_PASSES", "1"))


def disc_backward_lockstep(passes, lead=0):
    """Two or three independent backward passes of one structure (the adversarial and the restoration pass over tape 1+2 and the first
    consistency pass over tape 3+4: a decoder -- the two decoders are mirrors --, heads and the trunk, different tapes and cotangents)
    advanced together: where all are about to issue the data gradient of the same layer, those go out as ONE launch
    (kernels.conv_group: same shape, two or three problems in one grid) -- on the 4x4 ... 1x1 levels a single pass's launches fill a
    fraction of the chip.  Everything else each pass issues is issued in that pass's own order, the first pass's first; every buffer two
    passes add into sees them in the order of the list.  Results equal the passes run one after the other up to the group launches'
    grouping of their K sums (their raw weight gradients go to separate temps: lane).  passes: list of (args, keywords) of
    disc_backward.  Returns the passes' input gradients."""
    gens, res = [None] * len(passes), [None] * len(passes)

    def start(i):          # (a pass may be given as a callable: its arguments are then built when it starts -- after `lead`, see below)
        a, kw = passes[i]() if callable(passes[i]) else passes[i]
        gens[i] = _disc_backward_gen(*a, **dict(kw, lane=i))

    def step(i):
        try:
            return next(gens[i])
        except StopIteration as e:
            res[i] = e.value
            return None
    # lead: the FIRST pass issues its first `lead` hand-outs by itself before the other passes start -- they may read what it leaves
    # behind (the restoration pass's decoder cotangents, which the second consistency pass sums into its decoder weight gradients)
    start(0)
    first = step(0)
    for _ in range(lead):
        if first is None:
            break
        K.conv(*first[0], **first[1])
        first = step(0)
    for i in range(1, len(passes)):
        start(i)
    cur = [first] + [step(i) for i in range(1, len(passes))]
    while any(c is not None for c in cur):
        live = [i for i, c in enumerate(cur) if c is not None]
        if len(live) >= 2:
            K.conv_group([cur[i] for i in live])      # (single launches where the group does not qualify)
        else:
            K.conv(*cur[live[0]][0], **cur[live[0]][1])
        for i in live:
            cur[i] = step(i)
    return res


def _disc_backward_gen(rt, P, tp, g_enc, g_dec, g_rec, sink, want_input_grad, dec_export=None, dec_import=None, flush=None,
                       overwrite=frozenset(), lane=0):
    """A generator: it YIELDS the data-gradient launches of the 3x3 layers as (args, keywords) of kernels.conv instead of issuing them
    (disc_backward issues each at once; disc_backward_lockstep pairs them with another pass's) and issues everything else itself.
    lane: which set of raw weight-gradient temps the pass uses (two passes advanced together must not share them).
    Replay one recorded pass.  g_enc (B,1,1,1) / g_dec (B,64,64,1) / g_rec (B,64,64,1): output
    cotangents (any may be None).  sink: GradSink or None.  All parameter gradients are ACCUMULATED.
    The decoders are task-specific: their parameter gradient is the SUM over the tasks that reach them through this tape,
    and a weight gradient is linear in the cotangent -- so a pass may hand its decoder cotangents over instead of computing
    the decoder weight gradients (dec_export: dict to fill, layer -> cotangent tensor) and the last pass over the tape
    computes them once from the sums (dec_import: the dicts of the earlier passes merged).
    overwrite: names of spectral-norm layers whose corrected weight gradient REPLACES what the sink holds instead of being added
    to it (the first pass into a gradient buffer: the buffer then needs no zero fill and is not read).
    flush: optional callable(stage) for a caller that ships finished gradients while the pass is still running (data
    parallelism: the all-reduce of a slice overlaps the rest of this pass).  Called with "heads" when every decoder / head
    gradient of this pass has been enqueued, spectral-norm corrections included, and with "trunk_low" when the same holds
    for the bottleneck and trunk levels 6 .. FLUSH_LEVEL; the work it waits for is on the side stream (side.run orders after it)."""
    B = tp.B
    x = tp.x_in
    dev = x.device
    sn_touched = []
    sn_act = {}                 # layer -> (cotangent, second cotangent or None, saved output, 1 / slope): see wgrad_sn
    sn_pre = set()              # layers whose two halves went through one launch (the temp holds the gradient over both sigmas)
    side = K.side_stream(dev)   # weight gradients run beside the data-gradient chain

    def want(name):
        return sink is not None and sink.get(name) is not None

    Bh = getattr(tp, "pair", 0)
    in_decoder = [False]

    def cot(key, p):
        """The cotangent a decoder weight gradient is taken from: None if it is handed over to a later pass, the sum with
        the earlier passes' cotangents if this pass computes it for them, p otherwise (trunk and heads of the encoder)."""
        if not in_decoder[0]:
            return lambda: p
        if dec_export is not None:
            dec_export[key] = p
            return None
        if dec_import is not None and key in dec_import:
            other = dec_import[key]
            f = lambda: K.add(p, other)             # evaluated on the side stream, right before the weight gradient
            f.parts = (p, other)                    # (a launch that adds the two as it loads them takes these instead)
            return f
        return lambda: p

    def wgrad_sn(name, p, q, gspec, N, Cc, k, yout=None, slope=0.2):
        """Raw weight gradient of an SN layer into the pass's temp (corrected and accumulated by mtd_sn_grad below).
        gspec = (h, k, stride, pad) of the forward conv.  A paired tape needs the two passes' gradients separately (each
        has its own sigma, u, v), so its batch halves go through two launches.
        yout: the layer's saved output (after its activation of negative slope `slope`; 1.0: none) -- where the layer has fewer
        output pixels than weight columns, the correction's <G, W> is taken from (p, yout) instead of from the gradient and the
        weights (mtd_sn_grad_layer.act_*)."""
        wn, bn = name + ".weight_orig", name + ".bias"
        if want(wn):
            src = cot(name, p)
            if src is None:
                return
            if SN_ACT_DOT and yout is not None and p.shape[0] * p.shape[1] * p.shape[2] < Cc * k * k and N % 4 == 0:
                pe, pe2 = getattr(src, "parts", None) or (p, None)
                if (all(t is None or (t.data_ptr() % 16 == 0 and K.ld_of(t) % 4 == 0) for t in (pe, pe2, yout))
                        and P[name + ".bias"].data_ptr() % 16 == 0):
                    sn_act[name] = (pe, pe2, yout, 1.0 / slope)      # (references: the cotangents stay allocated until sn_fix has read them)
            hh, kk, ss, pp = gspec
            if Bh:
                gfull = K.geom_fwd(B, hh, hh, kk, ss, pp)
                m_first = Bh * p.shape[1] * p.shape[2]
                if SN_MERGE_HALVES and name in sn_act and m_first % 32 == 0 and K.wgrad_half_ok(gfull, N, Cc, m_first):
                    # both halves in ONE launch of the small-map kernels: each half's cotangent times its own 1 / sigma as it is used, so
                    # the temp holds G_1 / sigma_1 + G_2 / sigma_2 and the correction adds two rank-one terms (its dot products come from
                    # the activation side: the raw gradients do not exist apart any more)
                    i = SN_INDEX[name]
                    half = (tp.sig[i, 1:2], tp.sig2[i, 1:2], m_first)
                    side.run(lambda: K.wgrad(src(), q, gfull, N, Cc, rt.gtemp(name, dev, 2 * lane), Cc * k * k, k * k, db=sink.get(bn),
                                             accumulate=False, accumulate_bias=True, half=half), p, q)
                    sn_pre.add(name)
                    sn_touched.append(name)
                    return

                def both():         # (one launch for both halves where the library's plan allows it)
                    pe, pe2 = getattr(src, "parts", None) or (src(), None)
                    K.wgrad_pair(pe, q, gfull, Bh, N, Cc, rt.gtemp(name, dev, 2 * lane), rt.gtemp(name, dev, 2 * lane + 1), Cc * k * k, k * k,
                                 db=sink.get(bn), accumulate_bias=True, p_add=pe2)
                side.run(both, p, q)
            else:
                geom = K.geom_fwd(B, hh, hh, kk, ss, pp)
                side.run(lambda: K.wgrad(src(), q, geom, N, Cc, rt.gtemp(name, dev, 2 * lane), Cc * k * k, k * k, db=sink.get(bn), accumulate=False,
                                         accumulate_bias=True), p, q)
            sn_touched.append(name)
        elif want(bn):
            raise NotImplementedError("bias-only gradient request")

    def dgrad_s1_call(name, gpre, r, N, Cc, out, k=3, mask=None, add1=None):
        # data gradient of a stride-1 SN conv with N input channels (outputs of this launch), Cc output channels: (args, keywords)
        gd = K.geom_dgrad_s1(B, r, r, k, (k - 1) // 2)
        return ((gpre, P[name + ".weight_orig"], gd, N, Cc, k * k, N * k * k, out),
                dict(add1=add1, mask=mask, mask_slope=0.2, **_scales(tp, name, gd)))


    def sn_fix():
        """Corrects and accumulates the raw weight gradients taken since the last call (one launch, on the side stream:
        the stream of the weight gradients it corrects; callers join() before reading the sink)."""
        if not sn_touched:
            return
        L = _lib.lib()
        structs, held = [], []
        for name in sn_touched:
            i = SN_INDEX[name]
            s = _lib.SnGradLayer()
            s.G = rt.gtemp(name, dev, 2 * lane).data_ptr()
            s.w = P[name + ".weight_orig"].data_ptr()
            s.u = tp.u_save.data_ptr() + 4 * SN_ROW_OFF[i]
            s.v = tp.v_save.data_ptr() + 4 * SN_COL_OFF[i]
            s.sigma = tp.sig.data_ptr() + 8 * i
            s.g_out = sink.get(name + ".weight_orig").data_ptr()
            s.rows, s.cols, s.accumulate = SN_SPECS[i][1], SN_SPECS[i][2], 0 if name in overwrite else 1
            if Bh:      # the second half's own sigma, u, v: corrected and added in the same launch, after the first
                if name in sn_pre:
                    s.prescaled = 1                  # (one gradient, already over both sigmas: wgrad_sn)
                else:
                    s.G2 = rt.gtemp(name, dev, 2 * lane + 1).data_ptr()
                s.u2 = tp.u_save2.data_ptr() + 4 * SN_ROW_OFF[i]
                s.v2 = tp.v_save2.data_ptr() + 4 * SN_COL_OFF[i]
                s.sigma2 = tp.sig2.data_ptr() + 8 * i
            act = sn_act.pop(name, None)
            if act is not None:
                pe, pe2, yout, inv_slope = act
                if True:
                    npix = pe.shape[0] * pe.shape[1] * pe.shape[2]
                    s.act_gy, s.act_gy_ld = pe.data_ptr(), K.ld_of(pe)
                    if pe2 is not None:
                        s.act_gy2, s.act_gy2_ld = pe2.data_ptr(), K.ld_of(pe2)
                    s.act_a, s.act_a_ld = yout.data_ptr(), K.ld_of(yout)
                    s.act_bias = P[name + ".bias"].data_ptr()
                    s.act_M, s.act_M_first, s.act_inv_slope = npix, (Bh * (npix // B) if Bh else npix), inv_slope
                    held.extend(t for t in (pe, pe2, yout) if t is not None)
            structs.append(s)
        del sn_touched[:]
        sn_pre.clear()
        def fix():
            # (the descriptor table is built HERE, under the side stream: a new table's upload is then ordered before its reader
            # by the stream itself)
            dev_tab, host_arr = K.device_table(structs, dev)
            need = L.mtd_sn_grad_ws_bytes(C.cast(host_arr, C.c_void_p), len(structs))
            ws = K.workspace(need, dev)
            K.check(L.mtd_sn_grad(dev_tab.data_ptr(), C.cast(host_arr, C.c_void_p), len(structs), ws.data_ptr(), K.stream_ptr()), "mtd_sn_grad")
            K.crosses_streams(*held)             # (cotangents / activations of the main stream read here, on the side stream)
        # (its operands are the raw weight gradients the side stream itself produced, the layer's weights and the saved u / v / sigma of
        # the forward pass: nothing the main stream has enqueued since the last fork -- no new hand-off)
        side.run(fix, fork=False)

    def flush_point(stage):
        if flush is not None and stage in FLUSH_STAGES:
            sn_fix()
            flush(stage)

    g_bot_parts = []
    g_skip = {l: [] for l in range(1, 7)}
    g1 = K.geom_fwd(B, 1, 1, 1, 1, 0)
    g64 = K.geom_fwd(B, 64, 64, 1, 1, 0)

    def decoder_backward(pre, g_out, cats, o1s, o2s, ins, head):
        """shared by the SEG ('s') and REC ('r') decoders; returns the gradient of x_bot (a generator like its caller: the data
        gradients of its 3x3 layers are handed out)"""
        t6 = o2s[6]
        in_decoder[0] = True
        if want(head + ".weight"):
            src_h = cot(head, g_out)
            if src_h is not None:
                side.run(lambda: K.wgrad(src_h(), t6, g64, 1, 1, sink.get(head + ".weight"), 1, 1, db=sink.get(head + ".bias"), accumulate=True), g_out, t6)
        # every cotangent that reaches a level's second conv is multiplied by that conv's LeakyReLU gradient; the producer of the
        # cotangent (head conv, upsampling adjoint, 1x1 up-conv data gradient) applies it in its epilogue instead of a pass of its own
        g = K.empty_nhwc(B, 64, 64, 1, x)
        K.conv(g_out, P[head + ".weight"], g64, 1, 1, 1, 1, g, mask=o2s[6], mask_slope=0.2)
        for lvl in range(6, 0, -1):
            r = 2 ** lvl
            ccat, co = DEC[lvl - 1]
            o2, o1, cat, tin = o2s[lvl], o1s[lvl], cats[lvl], ins[lvl]
            g3 = K.geom_fwd(B, r, r, 3, 1, 1)
            gpre2 = g                                      # already times (o2 > 0 ? 1 : 0.2)
            below = o2s[lvl - 1] if lvl > 1 else None      # output of the level below (None: the bottleneck, masked by the caller)
            wgrad_sn(f"{pre}_dconv{lvl}2", gpre2, o1, (r, 3, 1, 1), co, co, 3, yout=o2)
            gpre1 = K.empty_nhwc(B, r, r, co, x)
            yield dgrad_s1_call(f"{pre}_dconv{lvl}2", gpre2, r, co, co, gpre1, mask=o1)
            wgrad_sn(f"{pre}_dconv{lvl}1", gpre1, cat, (r, 3, 1, 1), co, ccat, 3, yout=o1)
            gcat = K.empty_nhwc(B, r, r, ccat, x)
            yield dgrad_s1_call(f"{pre}_dconv{lvl}1", gpre1, r, ccat, co, gcat)
            cprev = tin.shape[3] if pre == "s" else RUP[lvl - 1][1]
            g_skip[7 - lvl].append(gcat[..., cprev:])
            if pre == "s":
                g = K.empty_nhwc(B, r // 2, r // 2, cprev, x)
                K.upsample2x_bwd(gcat[..., :cprev], g, mask=below, slope=0.2)
            else:
                cin_up, cup = RUP[lvl - 1]
                gr = K.empty_nhwc(B, r // 2, r // 2, 4 * cup, x)
                K.pixel_shuffle2_bwd(gcat[..., :cup], gr)
                gq = K.geom_fwd(B, r // 2, r // 2, 1, 1, 0)
                wn = f"r_up{lvl}.upsample.0.weight"
                if want(wn):
                    src_u = cot(f"r_up{lvl}", gr)
                    if src_u is not None:
                        side.run(lambda src_u=src_u, tin=tin, gq=gq, wn=wn, cup=cup, cin_up=cin_up, lvl=lvl: K.wgrad(
                            src_u(), tin, gq, 4 * cup, cin_up, sink.get(wn), cin_up, 1, db=sink.get(f"r_up{lvl}.upsample.0.bias"), accumulate=True), gr, tin)
                g = K.empty_nhwc(B, r // 2, r // 2, cin_up, x)
                K.conv(gr, P[wn], gq, cin_up, 4 * cup, 1, cin_up, g, mask=below, mask_slope=0.2)
        in_decoder[0] = False
        return g

    def decoders_backward_both():
        """decoder_backward for the restoration and the pixel-level decoder level by level, the data gradients of their mirror convs in
        pairs (kernels.conv_pair).  Every launch and every append happens in the order of two decoder_backward calls, "r" first."""
        in_decoder[0] = True
        D2 = {"r": (g_rec, tp.r_cat, tp.r_o1, tp.r_o2, tp.r_in, "rec_out"), "s": (g_dec, tp.s_cat, tp.s_o1, tp.s_o2, tp.s_in, "dec_out")}
        g = {}
        for pre in "rs":
            g_out, _cats, _o1s, o2s, _ins, head = D2[pre]
            t6 = o2s[6]
            if want(head + ".weight"):
                src_h = cot(head, g_out)
                if src_h is not None:
                    side.run(lambda src_h=src_h, t6=t6, head=head: K.wgrad(src_h(), t6, g64, 1, 1, sink.get(head + ".weight"), 1, 1,
                                                                          db=sink.get(head + ".bias"), accumulate=True), g_out, t6)
            g[pre] = K.empty_nhwc(B, 64, 64, 1, x)
            K.conv(g_out, P[head + ".weight"], g64, 1, 1, 1, 1, g[pre], mask=o2s[6], mask_slope=0.2)
        for lvl in range(6, 0, -1):
            r = 2 ** lvl
            ccat, co = DEC[lvl - 1]
            gpre1, gcat = {}, {}
            for pre in "rs":
                wgrad_sn(f"{pre}_dconv{lvl}2", g[pre], D2[pre][2][lvl], (r, 3, 1, 1), co, co, 3, yout=D2[pre][3][lvl])
                gpre1[pre] = K.empty_nhwc(B, r, r, co, x)
            K.conv_pair(*[dgrad_s1_call(f"{pre}_dconv{lvl}2", g[pre], r, co, co, gpre1[pre], mask=D2[pre][2][lvl]) for pre in "rs"])
            for pre in "rs":
                wgrad_sn(f"{pre}_dconv{lvl}1", gpre1[pre], D2[pre][1][lvl], (r, 3, 1, 1), co, ccat, 3, yout=D2[pre][2][lvl])
                gcat[pre] = K.empty_nhwc(B, r, r, ccat, x)
            K.conv_pair(*[dgrad_s1_call(f"{pre}_dconv{lvl}1", gpre1[pre], r, ccat, co, gcat[pre]) for pre in "rs"])
            for pre in "rs":
                _g_out, _cats, _o1s, o2s, ins, _head = D2[pre]
                tin = ins[lvl]
                below = o2s[lvl - 1] if lvl > 1 else None
                cprev = tin.shape[3] if pre == "s" else RUP[lvl - 1][1]
                g_skip[7 - lvl].append(gcat[pre][..., cprev:])
                if pre == "s":
                    g[pre] = K.empty_nhwc(B, r // 2, r // 2, cprev, x)
                    K.upsample2x_bwd(gcat[pre][..., :cprev], g[pre], mask=below, slope=0.2)
                else:
                    cin_up, cup = RUP[lvl - 1]
                    gr = K.empty_nhwc(B, r // 2, r // 2, 4 * cup, x)
                    K.pixel_shuffle2_bwd(gcat[pre][..., :cup], gr)
                    gq = K.geom_fwd(B, r // 2, r // 2, 1, 1, 0)
                    wn = f"r_up{lvl}.upsample.0.weight"
                    if want(wn):
                        src_u = cot(f"r_up{lvl}", gr)
                        if src_u is not None:
                            side.run(lambda src_u=src_u, tin=tin, gq=gq, wn=wn, cup=cup, cin_up=cin_up, lvl=lvl: K.wgrad(
                                src_u(), tin, gq, 4 * cup, cin_up, sink.get(wn), cin_up, 1, db=sink.get(f"r_up{lvl}.upsample.0.bias"), accumulate=True), gr, tin)
                    g[pre] = K.empty_nhwc(B, r // 2, r // 2, cin_up, x)
                    K.conv(gr, P[wn], gq, cin_up, 4 * cup, 1, cin_up, g[pre], mask=below, mask_slope=0.2)
        in_decoder[0] = False
        return g["r"], g["s"]

    if PAIR_DECODERS and g_rec is not None and g_dec is not None:
        g_bot_parts.extend(decoders_backward_both())
    else:
        if g_rec is not None:
            g_bot_parts.append((yield from decoder_backward("r", g_rec, tp.r_cat, tp.r_o1, tp.r_o2, tp.r_in, "rec_out")))
        if g_dec is not None:
            g_bot_parts.append((yield from decoder_backward("s", g_dec, tp.s_cat, tp.s_o1, tp.s_o2, tp.s_in, "dec_out")))
    if g_enc is not None:
        if want("enc_out.weight"):
            side.run(lambda: K.wgrad(g_enc, tp.cm, g1, 1, 512, sink.get("enc_out.weight"), 512, 1, db=sink.get("enc_out.bias"), accumulate=True), g_enc)
        gcm = K.empty_nhwc(B, 1, 1, 512, x)
        K.conv(g_enc, P["enc_out.weight"], g1, 512, 1, 1, 512, gcm)
        gc = K.mul(gcm, tp.drop_mask.reshape(B, 1, 1, 512)) if tp.drop_mask is not None else gcm
        gpre = K.act_grad(gc, tp.c, 0.2)
        wn = "c_fc.weight_orig"
        if want(wn):
            wgrad_sn("c_fc", gpre, tp.bot, (1, 1, 1, 0), 512, 512, 1, yout=tp.c)
        gb = K.empty_nhwc(B, 1, 1, 512, x)
        K.conv(gpre, P[wn], g1, 512, 512, 1, 512, gb, **_scales(tp, "c_fc", g1))
        g_bot_parts.append(gb)

    flush_point("heads")

    # ---- bottleneck
    gbot = g_bot_parts[0]
    for extra in g_bot_parts[1:]:
        K.copy_channels(extra, gbot, accumulate=True)
    gpre = K.act_grad(gbot, tp.bot, 0.2)
    wgrad_sn("bconv2", gpre, tp.b1, (1, 1, 1, 0), 512, 512, 1, yout=tp.bot)
    gpre1 = K.empty_nhwc(B, 1, 1, 512, x)
    K.conv(gpre, P["bconv2.weight_orig"], g1, 512, 512, 1, 512, gpre1, mask=tp.b1, mask_slope=0.2, **_scales(tp, "bconv2", g1))
    wgrad_sn("bconv1", gpre1, tp.d6, (1, 1, 1, 0), 512, 512, 1, yout=tp.b1)
    g = K.empty_nhwc(B, 1, 1, 512, x)
    K.conv(gpre1, P["bconv1.weight_orig"], g1, 512, 512, 1, 512, g, **_scales(tp, "bconv1", g1))

    # ---- trunk, levels 6..1
    g_in = None
    for l in range(6, 0, -1):
        h = 64 >> (l - 1)
        co = CH[l - 1]
        ci = 1 if l == 1 else CH[l - 2]
        xl, a, tin = tp.xs[l], tp.a[l], tp.tin[l]
        wgrad_sn(f"down{l}", g, xl, (h, 4, 2, 1), co, co, 4, yout=(tp.tin[l + 1] if l < 6 else tp.d6), slope=1.0)
        gpre2 = K.empty_nhwc(B, h, h, co, x)
        adds = g_skip[l]
        add1 = adds[0] if len(adds) > 0 else None
        add2 = adds[1] if len(adds) > 1 else None
        wd = P[f"down{l}.weight_orig"]
        # the four input-parity classes of the stride-2 data gradient: one grid (kernels.conv_multi)
        calls = []
        for py in range(2):
            for px in range(2):
                gp = K.geom_dgrad_s2(B, h, h, py, px)
                calls.append(((g, wd, gp, co, co, 16, co * 16, gpre2),
                              dict(add1=add1, add2=add2, mask=xl, mask_slope=0.2, **_scales(tp, f"down{l}", gp))))
        K.conv_multi(calls)
        g3 = K.geom_fwd(B, h, h, 3, 1, 1)
        wgrad_sn(f"conv{l}2", gpre2, a, (h, 3, 1, 1), co, co, 3, yout=xl)
        gpre1 = K.empty_nhwc(B, h, h, co, x)
        yield dgrad_s1_call(f"conv{l}2", gpre2, h, co, co, gpre1, mask=a)
        wgrad_sn(f"conv{l}1", gpre1, tin, (h, 3, 1, 1), co, ci, 3, yout=a)
        if l == FLUSH_LEVEL:
            flush_point("trunk_low")
        if l > 1:
            g = K.empty_nhwc(B, h, h, ci, x)
            yield dgrad_s1_call(f"conv{l}1", gpre1, h, ci, co, g)
        elif want_input_grad:
            g_in = K.empty_nhwc(B, h, h, 1, x)
            yield dgrad_s1_call("conv11", gpre1, h, 1, co, g_in)

    # ---- spectral-norm correction of the raw weight gradients, accumulated into the sink
    sn_fix()
    return g_in

"""Tensor-level launch helpers over the C ABI (include/mtdgan_hip.h).  PyTorch supplies device memory and
the current HIP stream; every arithmetic op below runs in libmtdgan_hip.so.  NHWC activations are torch
tensors of shape (B, H, W, C) whose last-dim stride is 1 and whose pixel stride (`ld`) may exceed C
(channel slices of a concat buffer)."""
from . import _options
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_RELU_ADD, ConvArgs, Geom, WgradArgs, check  # noqa: F401

_ws = {}
_pack_cache = {}
_t64_cache = {}     # transposed 64 x 64 mix weights of the Res-FFT blocks
_latest = {}        # (cache id, identity of a derived view without its version) -> current key: one generation per view
_pack_epoch = 0


_view_bufs = {}     # identity of a derived weight view -> its buffer.  A view is REWRITTEN IN PLACE when its weights change
# (same address for the life of the process), so a recorded launch list -- whose launches carry addresses -- can pack in one
# iteration what the next iteration's first launches read, exactly like consecutive eager iterations share their views.  Safe
# in stream order: views are packed on the main stream, after the joins that precede an optimizer step.


def _view_buffer(ident, shape, device):
    buf = _view_bufs.get(ident)
    if buf is None or buf.device != device or tuple(buf.shape) != tuple(shape):
        buf = torch.empty(shape, dtype=torch.float32, device=device)
        _view_bufs[ident] = buf
    return buf


def release_views(params):
    """Free the derived-view buffers (packed / Winograd-transformed / transposed weights) of these parameters -- for a caller
    that drops a model and keeps the process: the buffers are keyed by the weights' addresses and would otherwise stay for
    the life of the process.  Call it only when nothing will launch on the model again (a recorded launch list of that model
    carries the buffers' addresses)."""
    spans = []
    for p in params:
        st = p.untyped_storage()
        spans.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
    dead = [i for i in _view_bufs if any(lo <= i[0] < hi for lo, hi in spans)]
    for i in dead:
        del _view_bufs[i]
    weights_changed(None)
    return len(dead)


def _remember(cache, ident, key, value):
    """cache[key] = value, dropping the entry of an older version of the same view (an optimizer that updates in place,
    e.g. torch.optim.AdamW, bumps the version counter on every step; without this the old generations would pile up)."""
    old = _latest.get((id(cache), ident))
    if old is not None and old != key:
        cache.pop(old, None)
    _latest[(id(cache), ident)] = key
    cache[key] = value


def weights_changed(params=None):
    """Called by the optimizers after they update parameters through raw pointers (the tensors' version counters do
    not move): drops the packed views of those parameters (of every parameter when params is None) and bumps the
    `_mtd_epoch` stamp that caches of derived quantities (train_step: the generator tape) compare."""
    global _pack_epoch
    if params is None:
        _pack_epoch += 1
        _pack_cache.clear()
        _t64_cache.clear()
        _latest.clear()
        return
    stor = set()
    for p in params:
        stor.add(p.untyped_storage().data_ptr())
        p._mtd_epoch = getattr(p, "_mtd_epoch", 0) + 1
    for cache in (_pack_cache, _t64_cache):
        for key in [k for k, (_dst, w) in cache.items() if w.untyped_storage().data_ptr() in stor]:
            del cache[key]


def prepack(views):
    """Pack many weight views in ONE launch.  views: iterable of (w, N, C, w_sn, w_sc) exactly as conv() will
    ask for them; already cached / natively c-contiguous views are skipped."""
    todo = []
    for (w, N, Cc, w_sn, w_sc) in views:
        if (Cc % 32) or (N % 32) or not (w_sc != 1 or w.data_ptr() % 16 or w_sn % 4):
            continue
        key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc)
        if key in _pack_cache:
            continue
        T = w.numel() // (N * Cc)
        dst = _view_buffer((w.data_ptr(), N, Cc, w_sn, w_sc), (T, N, Cc), w.device)
        d = _lib.PackDesc()
        d.src, d.dst, d.N, d.C, d.T, d.sn, d.sc = w.data_ptr(), dst.data_ptr(), N, Cc, T, w_sn, w_sc
        todo.append(d)
        _remember(_pack_cache, (w.data_ptr(), N, Cc, w_sn, w_sc), key, (dst, w))
    if todo:
        tab, host = device_table(todo, todo and views[0][0].device)
        check(_lib.lib().mtd_pack_weights(tab.data_ptr(), C.cast(host, C.c_void_p), len(todo), stream_ptr()), "mtd_pack_weights")


def packed_weight_view(w, N, Cc, w_sn, w_sc):
    """[tap][n][c] copy of the weight view W(n,c,tap) = w[n*w_sn + c*w_sc + tap] (c contiguous, which is what
    the implicit-GEMM kernel stages with 16-byte loads).  Cached until the weights change."""
    T = w.numel() // (N * Cc)
    key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc)
    hit = _pack_cache.get(key)
    if hit is None:
        dst = _view_buffer((w.data_ptr(), N, Cc, w_sn, w_sc), (T, N, Cc), w.device)
        d = _lib.PackDesc()
        d.src, d.dst, d.N, d.C, d.T, d.sn, d.sc = w.data_ptr(), dst.data_ptr(), N, Cc, T, w_sn, w_sc
        tab, host = device_table([d], w.device)
        check(_lib.lib().mtd_pack_weights(tab.data_ptr(), C.cast(host, C.c_void_p), 1, stream_ptr()), "mtd_pack_weights")
        hit = (dst, w)          # keep the source alive so its data_ptr cannot be recycled under the same key
        _remember(_pack_cache, (w.data_ptr(), N, Cc, w_sn, w_sc), key, hit)
    return hit[0], Cc, 1, N * Cc


# ---- Winograd F(2x2, 3x3) for the 3x3 stride-1 layers (csrc/conv_winograd.hip).  MTD_WINOGRAD=0 switches it off;
# MTD_WINOGRAD_MIN_HW / _MIN_C / _MIN_N bound the layers that take it.  (Measured: even the 4 x 4 and 2 x 2 maps of the
# deepest levels gain -- 942 against 926 img/s with them -- although the implicit GEMM's whole-tile tap skipping already drops
# most of their padding taps: 512 x 512 on 4 x 4 maps 54 -> 35 us.)
if _options.lab("MTD_WINO_SPLIT", "") != "":       # lab: the split-bf16 Winograd kernel for a whole run (mtd_set_option("wino_split", ...))
    _lib.lib().mtd_set_option(b"wino_split", int(_options.lab("MTD_WINO_SPLIT", "0")))
WINOGRAD = _options.lab("MTD_WINOGRAD", "1") != "0"
WINO_MIN_HW = int(_options.lab("MTD_WINOGRAD_MIN_HW", "2"))
WINO_MIN_C = int(_options.lab("MTD_WINOGRAD_MIN_C", "64"))
WINO_MIN_N = int(_options.lab("MTD_WINOGRAD_MIN_N", "64"))
# The generator's 32 -> 32 layers on the F(2x4) kernel's 32-channel form (NB = 1, two K steps per launch tile): on maps of at
# least this side.  Default 128 = whole-slice inference only (512 x 512: 29.36 -> 27.99 ms per slice); on the 64 x 64 training
# patches the halo-tile kernel is as fast (generator leg 5.44 -> 5.39 ms with 64 here) and carries the fused epilogues
# (out2, block tail) the Winograd kernel does not have.  0 switches the form off.
WINO_C32_MIN_HW = int(_options.lab("MTD_WINO_C32_MIN_HW", "128"))
# ... and the FORWARD pass of the generator's plain encoder / decoder layers on the training patches too (conv(..., wino32=True):
# 23 us per layer against the halo-tile kernel's 28; generator leg 5.45 -> 5.33 ms).  The backward pass keeps its fused kernels.
WINO_C32_FWD = _options.lab("MTD_WINO_C32_FWD", "1") != "0"
# ... and (round 6) their DATA GRADIENTS: the kernel's MASKED2 form writes the cotangent and its masked form for the consuming block
# (conv(..., wino32=True, mask=..., out2=...)); the weight gradient goes to wgrad_wino32_kernel on the side stream
WINO_C32_BWD = _options.lab("MTD_WINO_C32_BWD", "1") != "0"
_kmap_cache = {}


def _wino_kmap(geom):
    key = bytes(geom)
    km = _kmap_cache.get(key)
    if km is None:
        arr = (C.c_int * 9)()
        rc = _lib.lib().mtd_winograd_kmap(C.byref(geom), arr)
        km = tuple(arr) if rc == 0 else ()
        _kmap_cache[key] = km
    return km


def winograd_takes(geom, N, Cc, kw):
    """Host-side mirror of mtd_conv_winograd_ok plus the size thresholds: does conv() send this launch to the Winograd kernel?"""
    if not WINOGRAD or geom.TH != 3 or geom.TW != 3 or geom.in_sy != 1 or geom.in_sx != 1:
        return False
    if geom.IH != geom.OH or geom.IW != geom.OW or (geom.OH & 1) or (geom.OW & 1) or geom.OH < WINO_MIN_HW or geom.OW < WINO_MIN_HW:
        return False
    if not (geom.out_sy == 1 and geom.out_sx == 1 and geom.out_oy == 0 and geom.out_ox == 0 and geom.OHF == geom.OH and geom.OWF == geom.OW):
        return False
    if (Cc % 16) or (kw.get("out2") is not None and not (WINO_C32_BWD and kw.get("wino32") and Cc == 32 and N == 32 and kw.get("mask") is not None)):
        return False
    if Cc < WINO_MIN_C or (N % 64) or N < WINO_MIN_N:
        # (the 32-channel form; the library checks that the layer's transform is F(2x4): mtd_conv_winograd_ok)
        if not (WINO_C32_MIN_HW and Cc == 32 and N == 32 and geom.OW % 4 == 0):
            return False
        if min(geom.OH, geom.OW) < WINO_C32_MIN_HW:
            # smaller maps: only where the caller asks for it (wino32=True: the generator's forward pass) and the persistent
            # kernel itself takes the launch (mirror of wino_c32_takes: one residual operand at most, no scale, no mask)
            if not (WINO_C32_FWD and kw.get("wino32") and min(geom.OH, geom.OW) >= 16 and (kw.get("mask") is None or WINO_C32_BWD)
                    and kw.get("add2") is None and kw.get("scale") is None and kw.get("scale2") is None):
                return False
    elif kw.get("act") == ACT_RELU_ADD:
        return False
    return len(_wino_kmap(geom)) == 9


_wino_px_cache = {}


def winograd_patch_w(geom, N, Cc):
    """The transform the library's plan wants along x for this layer (mtd_conv_winograd_patch_w): 6 = F(2x4, 3x3) -- 3 MFMA
    multiplications per output pixel and channel pair -- or 4 = F(2x2, 3x3) -- 4 of them; the direct form has 9.  Plus 16 when the
    layer runs on the split-bf16 kernel (csrc/conv_winograd_split.h: its weights are three bf16 planes)."""
    key = (bytes(geom), N, Cc)
    px = _wino_px_cache.get(key)
    if px is None:
        a = ConvArgs()
        a.g = geom
        a.inp = a.w = a.out = 16                   # (the query looks at shapes only; non-null placeholders)
        a.in_ld, a.C, a.N, a.out_ld = Cc, Cc, N, N
        px = _lib.lib().mtd_conv_winograd_patch_w(C.byref(a))
        _wino_px_cache[key] = px
    return px


def winograd_f4_min_w(min_w):
    """Tuning / test hook (mtd_conv_winograd_f4_min_w): narrowest map that takes F(2x4, 3x3); 0 = never.  Returns the old value."""
    old = _lib.lib().mtd_conv_winograd_f4_min_w(int(min_w))
    _wino_px_cache.clear()
    _igemm_ws_cache.clear()
    return old


def _wino_desc(w, N, Cc, w_sn, w_sc, kmap, device, px):
    # (px: patch width 4 / 6, + 16 for the split-bf16 form -- three bf16 planes, 6 bytes per transformed weight)
    dst = _view_buffer((w.data_ptr(), N, Cc, w_sn, w_sc, "wino", kmap, px), ((6 if px & 16 else 4) * (px & 15) * N * Cc,), device)
    d = _lib.WinoWeightDesc()
    d.src, d.dst, d.sn, d.sc, d.st, d.N, d.C, d.px = w.data_ptr(), dst.data_ptr(), w_sn, w_sc, 1, N, Cc, px
    for i, k in enumerate(kmap):
        d.kmap[i] = k
    return d, dst


def prepack_winograd(views):
    """Transform many weight views in ONE launch.  views: iterable of (w, N, C, w_sn, w_sc, geom[, conv keywords]) exactly as
    conv() will ask for them; cached views and views conv() would not send to the Winograd kernel are skipped."""
    todo = []
    dev = None
    for v in views:
        w, N, Cc, w_sn, w_sc, geom = v[:6]
        if not winograd_takes(geom, N, Cc, v[6] if len(v) > 6 else {}):       # (v[6]: the keywords conv() will be called with)
            continue
        kmap = _wino_kmap(geom)
        px = winograd_patch_w(geom, N, Cc)
        key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc, "wino", kmap, px)
        if key in _pack_cache:
            continue
        d, dst = _wino_desc(w, N, Cc, w_sn, w_sc, kmap, w.device, px)
        todo.append(d)
        dev = w.device
        _remember(_pack_cache, (w.data_ptr(), N, Cc, w_sn, w_sc, "wino", kmap, px), key, (dst, w))
    if todo:
        tab, host = device_table(todo, dev)
        check(_lib.lib().mtd_winograd_weights(tab.data_ptr(), C.cast(host, C.c_void_p), len(todo), stream_ptr()), "mtd_winograd_weights")


def winograd_weight_view(w, N, Cc, w_sn, w_sc, geom):
    """The transformed weights [xi][C/8][N][8] of the view W(n,c,tap) = w[n*w_sn + c*w_sc + tap] for this geometry's tap
    order, and the patch width px (6: F(2x4, 3x3), 4: F(2x2, 3x3)) they were built for -- the library's choice per layer.
    Cached until the weights change (like the packed [tap][n][c] views)."""
    kmap = _wino_kmap(geom)
    px = winograd_patch_w(geom, N, Cc)
    key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc, "wino", kmap, px)
    hit = _pack_cache.get(key)
    if hit is None:
        d, dst = _wino_desc(w, N, Cc, w_sn, w_sc, kmap, w.device, px)
        tab, host = device_table([d], w.device)
        check(_lib.lib().mtd_winograd_weights(tab.data_ptr(), C.cast(host, C.c_void_p), 1, stream_ptr()), "mtd_winograd_weights")
        hit = (dst, w)
        _remember(_pack_cache, (w.data_ptr(), N, Cc, w_sn, w_sc, "wino", kmap, px), key, hit)
    return hit[0], px


# ---- F(3x3, 2x2) for the 4x4 / stride-2 layers (csrc/conv_wino_s2.h): the forward conv and the parity classes of its data gradient
WINO_S2 = int(_options.lab("MTD_WINO_S2", "1"))      # 0: never; 1: where the library's plan expects it to pay (mtd_conv_winograd_s2_ok == 2); 2: wherever it can
WINO_S2_MIN_HW = int(_options.lab("MTD_WINO_S2_MIN_HW", "8"))      # smallest output map (of a set) that takes it
_wino_s2_kmap_cache = {}


def _wino_s2_kmap(geom):
    key = bytes(geom)
    hit = _wino_s2_kmap_cache.get(key)
    if hit is None:
        groups = C.c_int(0)
        km = (C.c_int * 16)()
        rc = _lib.lib().mtd_winograd_s2_kmap(C.byref(geom), C.byref(groups), km)
        hit = (groups.value, tuple(km)) if rc == 0 else (0, ())
        _wino_s2_kmap_cache[key] = hit
    return hit


def winograd_s2_takes(geom, N, Cc, kw):
    """Host-side mirror of mtd_conv_winograd_s2_ok plus the size threshold."""
    if not WINO_S2 or (N % 64) or (Cc % 16) or kw.get("out2") is not None or kw.get("act") == ACT_RELU_ADD:
        return False
    if not ((geom.TH == 4 and geom.TW == 4 and geom.in_sy == 2 and geom.in_sx == 2) or (geom.TH == 2 and geom.TW == 2 and geom.in_sy == 1 and geom.in_sx == 1)):
        return False
    if min(geom.OH, geom.OW) < WINO_S2_MIN_HW:
        return False
    return _wino_s2_kmap(geom)[0] > 0


_wino_s2_demand = set()      # the views conv() / conv_multi() have asked for (prepack_winograd_s2 transforms these, and only these, ahead)


def _wino_s2_desc(w, N, Cc, w_sn, w_sc, groups, kmap):
    skey = (w.data_ptr(), N, Cc, w_sn, w_sc, "wino_s2", kmap, groups)
    dst = _view_buffer(skey, (16 * groups * N * Cc,), w.device)
    d = _lib.WinoS2WeightDesc()
    d.src, d.dst, d.sn, d.sc, d.st, d.N, d.C, d.groups = w.data_ptr(), dst.data_ptr(), w_sn, w_sc, 1, N, Cc, groups
    for i, k in enumerate(kmap):
        d.kmap[i] = k
    return skey, d, dst


def winograd_s2_weight_view(w, N, Cc, w_sn, w_sc, geom):
    """The transformed weights [xi][groups C / 8][N][8] of the view W(n, c, tap) for this geometry (mtd_winograd_s2_weights);
    cached until the weights change."""
    groups, kmap = _wino_s2_kmap(geom)
    key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc, "wino_s2", kmap, groups)
    hit = _pack_cache.get(key)
    if hit is None:
        skey, d, dst = _wino_s2_desc(w, N, Cc, w_sn, w_sc, groups, kmap)
        _wino_s2_demand.add(skey)
        tab, host = device_table([d], w.device)
        check(_lib.lib().mtd_winograd_s2_weights(tab.data_ptr(), C.cast(host, C.c_void_p), 1, stream_ptr()), "mtd_winograd_s2_weights")
        hit = (dst, w)
        _remember(_pack_cache, skey, key, hit)
    return hit[0]


def prepack_winograd_s2(views):
    """Transform, in ONE launch, those of `views` -- (w, N, C, w_sn, w_sc, geom) as conv() / conv_multi() will ask for them -- that
    an earlier step did ask for (whether a 4x4 stride-2 launch takes the F(3x3, 2x2) form is the library's plan for its batch and
    shape, so the first step finds out and the later ones prepare exactly those) and whose weights have changed since."""
    todo = []
    dev = None
    for (w, N, Cc, w_sn, w_sc, geom) in views:
        groups, kmap = _wino_s2_kmap(geom)
        if not groups or (w.data_ptr(), N, Cc, w_sn, w_sc, "wino_s2", kmap, groups) not in _wino_s2_demand:
            continue
        key = (w.data_ptr(), w._version, _pack_epoch, N, Cc, w_sn, w_sc, "wino_s2", kmap, groups)
        if key in _pack_cache:
            continue
        skey, d, dst = _wino_s2_desc(w, N, Cc, w_sn, w_sc, groups, kmap)
        todo.append(d)
        dev = w.device
        _remember(_pack_cache, skey, key, (dst, w))
    if todo:
        tab, host = device_table(todo, dev)
        check(_lib.lib().mtd_winograd_s2_weights(tab.data_ptr(), C.cast(host, C.c_void_p), len(todo), stream_ptr()), "mtd_winograd_s2_weights")


def _conv_winograd_s2(calls):
    """calls: [(conv() argument tuple, keywords)] of one shape (one forward conv, or the parity classes of a data gradient).
    True if the launch was made."""
    L = _lib.lib()
    arr = (ConvArgs * len(calls))(*[_conv_args(*args, pack=False, **kw) for args, kw in calls])
    if L.mtd_conv_winograd_s2_ok(arr, len(calls)) < (1 if WINO_S2 >= 2 else 2):
        if FLOP_COUNT is not None:
            for args, _ in calls:
                g = args[2]
                FLOP_COUNT["conv_mfma"] -= 2.0 * g.B * g.OH * g.OW * args[3] * args[4] * g.TH * g.TW
                FLOP_COUNT["launches"] -= 1
        return False
    for i, (args, _) in enumerate(calls):
        x, w, geom, N, Cc, w_sn, w_sc = args[:7]
        arr[i].w = winograd_s2_weight_view(w, N, Cc, w_sn, w_sc, geom).data_ptr()
    x, _, geom, N, Cc = calls[0][0][:5]
    key = ("wino_s2", bytes(geom), N, Cc, len(calls))
    need = _igemm_ws_cache.get(key)
    if need is None:
        need = L.mtd_conv_winograd_s2_ws_bytes(arr, len(calls))
        _igemm_ws_cache[key] = need
    if need:
        need = (need + 255) & ~255
        ws = workspace(need * len(calls), x.device)
        for i in range(len(calls)):
            arr[i].ws, arr[i].ws_bytes = ws.data_ptr() + i * need, need
    if FLOP_COUNT is not None:      # executed: 16 multiplications per 3 x 3 tile (ragged tiles in full) instead of taps per pixel
        for args, _ in calls:
            g = args[2]
            full = 2.0 * g.B * g.OH * g.OW * args[3] * args[4] * g.TH * g.TW
            groups = 4 if g.TH == 4 else 1
            done = 2.0 * g.B * ((g.OH + 2) // 3) * ((g.OW + 2) // 3) * 16 * args[3] * args[4] * groups
            FLOP_COUNT["conv_mfma"] -= full - done
            FLOP_COUNT["conv_winograd_saved"] = FLOP_COUNT.get("conv_winograd_saved", 0.0) + full - done
            if args is not calls[0][0]:
                FLOP_COUNT["launches"] -= 1
    check(L.mtd_conv_winograd_s2(arr, len(calls), stream_ptr()), "mtd_conv_winograd_s2")
    return True


def regrouped_bias(b, groups):
    """out[q * n + c] = b[c * groups + q]: the bias of a conv whose output channels are taken group by group (PixelShuffle
    classes, geom_pixel_shuffle2).  Cached like the packed weight views (dropped by weights_changed)."""
    key = (b.data_ptr(), b._version, _pack_epoch, "bias groups", groups)
    hit = _pack_cache.get(key)
    if hit is None:
        dst = _view_buffer((b.data_ptr(), "bias groups", groups), (b.numel(),), b.device)
        rec(dst.view(groups, -1).copy_, b.detach().view(-1, groups).t())
        hit = (dst, b)
        _remember(_pack_cache, (b.data_ptr(), "bias groups", groups), key, hit)
    return hit[0]


STATS = {"table_hit": 0, "table_miss": 0, "zero_copy_reads": 0}
# bench.py: while this is a dict, the helpers below add the dense algorithmic flop count of every launch they make
# (2*M*N*C*taps for convolutions and weight gradients, 2*64*64 per frequency for the spectral mix, 2.5*N*log2(N) per
# real 64 x 64 plane split evenly over the row and column passes) -- the work a step EXECUTES, as opposed to the
# reference-derived "sufficient" count its throughput is quoted against.
FLOP_COUNT = None
_FFT_HALF_PLANE = 2.5 * 4096 * 12 / 2


WGRAD_CFG_WINO = 16                            # mtd_conv_wgrad_plan_cfg: wgrad_wino_kernel (F(2x2, 3x3); 17 was its F(2x4) form, removed in round 6)
WGRAD_CFG_WINO_S2 = 18                         # wgrad_wino_s2_kernel: F(3x3, 2x2) over the four phases of a 4x4 / stride-2 layer
WGRAD_CFG_WINO32 = 19                          # wgrad_wino32_kernel: F(2x2, 3x3) on the generator's 32 x 32 block


def _count_wgrad(geom, N, Cc, cfg):
    """Executed flops of one weight-gradient launch: 2 M N C taps on the matrix cores (or the vector ALU for the degenerate
    channel counts); the Winograd kernels multiply 4 (F(2x2)) instead of 9 times per output pixel, the rest is
    `wgrad_winograd_saved`."""
    full = 2.0 * geom.B * geom.OH * geom.OW * N * Cc * geom.TH * geom.TW
    if cfg in (WGRAD_CFG_WINO, WGRAD_CFG_WINO_S2, WGRAD_CFG_WINO32):
        share = 4.0 / 9.0
        if cfg == WGRAD_CFG_WINO_S2:      # 16 multiplications per 3 x 3 tile (ragged tiles in full) and phase instead of 16 per pixel
            share = ((geom.OH + 2) // 3) * ((geom.OW + 2) // 3) * 16 * 4 / (geom.OH * geom.OW * 16.0)
        _count("wgrad_mfma", full * share)
        FLOP_COUNT["wgrad_winograd_saved"] = FLOP_COUNT.get("wgrad_winograd_saved", 0.0) + full * (1.0 - share)
    else:
        _count("wgrad_mfma" if (Cc % 32 == 0 and N % 32 == 0) else "wgrad_valu", full)


def _count(kind, flops):
    if FLOP_COUNT is not None:
        FLOP_COUNT[kind] = FLOP_COUNT.get(kind, 0.0) + float(flops)
        FLOP_COUNT["launches"] = FLOP_COUNT.get("launches", 0) + 1
_igemm_ws_cache = {}
CALL_LOG = None     # tools/tune_igemm.py: a list collects ("igemm" | "wgrad", bytes(argument struct)) per call

IGEMM_CONFIGS = ["igemm_kernel<2, 1, 4, 1>", "igemm_kernel<1, 1, 4, 1>", "igemm_kernel<2, 2, 4, 1>",
                 "igemm_kernel<1, 1, 2, 2>", "igemm_kernel<2, 2, 2, 2>", "igemm_kernel<1, 1, 1, 4>",
                 "igemm_tb_kernel<1>", "igemm_tb_kernel<2>", "igemm_v2_kernel<0>", "igemm_c32p_kernel", "igemm_c32t_kernel", "c32_bwd_kernel",
                 "igemm_c32t_kernel<4, true, true, true>", "c32_bwd_kernel<1, true>", "wino_conv_kernel<2, false, 4>", "wino_conv_kernel<4, false, 4>",      # 12: Res-FFT block tail; 13: c32_bwd + irfft; 14, 15, 22, 23: Winograd
                 "igemm_multi_kernel<2, 1, 4, 1>", "igemm_multi_kernel<1, 1, 4, 1>", "igemm_multi_kernel<2, 2, 4, 1>",      # 16 + cfg
                 "igemm_multi_kernel<1, 1, 2, 2>", "igemm_multi_kernel<2, 2, 2, 2>", "igemm_multi_kernel<1, 1, 1, 4>",
                 "wino_conv_kernel<2, true, 4>", "wino_conv_kernel<2, false, 6>", "wino_conv_kernel<1, false, 6>",             # 22, 23, 24 (6: F(2x4, 3x3))
                 "wino_c32_kernel<false, false>", "wino_c32_kernel<true, false>",                                               # 25, 26: the persistent 32 -> 32 channel form
                 "wino_conv3_kernel<6>", "wino_conv3_kernel<4>",                                                                # 27, 28: the split-bf16 forms (conv_winograd_split.h)
                 "wino32_conv_kernel<2, false>", "wino32_conv_kernel<2, true>", "wino32_conv_kernel<4, false>",
                 "wino_c32_kernel<false, true>", "wino_c32_kernel<true, true>",      # 32, 33: ... with a mask operand and a second output (round 6)
                 "wino_conv_multi_kernel<2, false, 6>", "wino_conv_multi_kernel<2, false, 4>", "wino_conv_multi_kernel<4, false, 4>",      # 34-37: two problems of
                 "wino_conv_multi_kernel<2, true, 4>"]                                                                                         # one shape per launch (round 6)                                                                                       # 29: F(3x3, 2x2) for the 4x4 / stride-2 layers (conv_wino_s2.h)
WGRAD_CONFIGS = ["wgrad_kernel<1, 1, 9>", "wgrad_kernel<1, 1, 4>", "wgrad_kernel<2, 2, 1>", "wgrad_kernel<1, 1, 8>",
                 "wgrad_kernel<1, 1, 3>", "wgrad_kernel<1, 1, 1>", "wgrad_kernel<2, 2, 3>",
                 "wgrad_row_kernel<3, 3, 1>", "wgrad_row_kernel<3, 3, -1>", "wgrad_row_kernel<1, 1, 1>",
                 "wgrad_blk_kernel<8>", "wgrad_blk_kernel<4>", "wgrad_blk_kernel<2>", "wgrad_taps_kernel", "?", "wgrad_s2_kernel",
                 "wgrad_wino_kernel", "?", "wgrad_wino_s2_kernel", "wgrad_wino32_kernel"]


SPECTRAL_KERNELS = ["rfft_rows_any_kernel", "spec_mix_any_kernel", "irfft_rows_any_kernel"]      # profiler class 2 (HBM-bound)


def prof_enable(capacity):
    """Switch the library's launch profiler on (HIP events around every igemm / wgrad main kernel) or off (0)."""
    L = _lib.lib()
    L.mtd_prof_enable.argtypes = [C.c_int]
    check(L.mtd_prof_enable(int(capacity)), "mtd_prof_enable")


def prof_mode(attach=-1):
    """Timing mode of the launch profiler (mtd_prof_mode): 1 = the kernel dispatch's own begin / end timestamps, 0 = events
    recorded before / after the launch; -1 queries."""
    return _lib.lib().mtd_prof_mode(int(attach))


def prof_collect(capacity):
    L = _lib.lib()
    buf = (_lib.ProfRecord * capacity)()
    L.mtd_prof_collect.argtypes = [C.c_void_p, C.c_int]
    n = L.mtd_prof_collect(C.cast(buf, C.c_void_p), capacity)
    out = []
    for r in buf[:min(n, capacity)]:
        names = (IGEMM_CONFIGS, WGRAD_CONFIGS, SPECTRAL_KERNELS)[r.kernel] if 0 <= r.kernel <= 2 else []
        out.append({"kernel": names[r.cfg] if 0 <= r.cfg < len(names) else "?", "splitk": r.splitk, "M": r.M, "N": r.N,
                    "C": r.C, "taps": r.taps, "flops": r.flops, "ms": r.ms, "bytes": r.bytes})
    return out


def igemm_override(cfg, splitk=1):
    """Tuning / diagnostic hook (mtd_conv_igemm_override): force an implicit-GEMM instantiation; (-1, -1) restores the plan."""
    L = _lib.lib()
    L.mtd_conv_igemm_override.argtypes = [C.c_int, C.c_int]
    L.mtd_conv_igemm_override(cfg, splitk)
    _igemm_ws_cache.clear()


if _options.lab("MTD_IGEMM_CFG", ""):        # diagnostic switch, e.g. MTD_IGEMM_CFG=9: persistent kernel for the 32-channel 3x3 layers
    igemm_override(int(_options.lab("MTD_IGEMM_CFG", "0")), 1)

_raw_stream = torch._C._cuda_getCurrentRawStream       # (device index) -> hipStream_t as int; no Python Stream objects
_cur_device = torch._C._cuda_getDevice


def stream_ptr():
    return C.c_void_p(_raw_stream(_cur_device()))


DEFER_WGRADS = _options.lab("MTD_NO_DEFERRED_WGRAD", "0") != "1"     # generator backward: slab sums of all layers in two launches
RECORDING = None    # the LaunchList being recorded, if any
CAPTURE_TAG = 0     # bumped while a hipGraph is captured so that graph-pool scratch never mixes with eager scratch


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream).  Stream order makes reuse safe."""
    idx = device.index if device.index is not None else _cur_device()
    key = (idx, _raw_stream(idx), CAPTURE_TAG)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# Split-K launches may finish inside the kernel (mtd_conv_args.tile_ctr: the last slice to arrive at a tile sums the slabs).
# Same bits as the separate epilogue launch; measured neutral on the full step (42.54 vs 42.42 ms), so off unless asked for.
SPLITK_FIN = _options.lab("MTD_SPLITK_FIN", "0") == "1"
TILE_CTRS = 4096        # arrival counters per set of a split-K launch (mtd_conv_args.tile_ctr); four sets per buffer
_tile_ctrs = {}


def tile_counters(device):
    """Zero-initialised arrival counters for the split-K launches of the current stream (they leave them zero, and
    launches of one stream do not overlap): one buffer per (device, stream), like workspace()."""
    idx = device.index if device.index is not None else _cur_device()
    key = (idx, _raw_stream(idx), CAPTURE_TAG)
    buf = _tile_ctrs.get(key)
    if buf is None:
        buf = torch.zeros(4 * TILE_CTRS, dtype=torch.int32, device=device)
        _tile_ctrs[key] = buf
    return buf


def _chk_nhwc(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.stride(3) == 1):
        raise ValueError(f"{name}: expected a CUDA fp32 NHWC tensor with unit channel stride, got {tuple(t.shape)} {t.stride()} {t.dtype} {t.device}")
    B, H, W, _ = t.shape
    ld = t.stride(2)
    if (H > 1 and t.stride(1) != W * ld) or (B > 1 and t.stride(0) != H * W * ld):
        if not (W == 1 and H == 1):
            raise ValueError(f"{name}: pixels are not uniformly strided: {tuple(t.shape)} {t.stride()}")
    return ld if (W > 1 or H > 1 or B > 1) else max(ld, t.shape[3])


def ld_of(t):
    """pixel stride of an NHWC tensor (for 1x1 images fall back to the batch stride)."""
    B, H, W, Cc = t.shape
    if W > 1:
        return t.stride(2)
    if H > 1:
        return t.stride(1)
    if B > 1:
        return t.stride(0)
    return Cc


def empty_nhwc(B, H, W, Cc, like):
    return torch.empty((B, H, W, Cc), dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------- geometry
def geom_fwd(B, IH, IW, k, s, p):
    OH, OW = (IH + 2 * p - k) // s + 1, (IW + 2 * p - k) // s + 1
    return Geom(B, IH, IW, OH, OW, s, s, -p, -p, 1, 1, k, k, k, 0, 0, 1, 1, OH, OW, 1, 1, 0, 0)


def geom_dgrad_s1(B, H, W, k, p):
    """stride-1 conv with input (H, W): gathers the conv OUTPUT-sized tensor at iy = y + p - ky."""
    GH, GW = H + 2 * p - k + 1, W + 2 * p - k + 1
    return Geom(B, GH, GW, H, W, 1, 1, p, p, -1, -1, k, k, k, 0, 0, 1, 1, H, W, 1, 1, 0, 0)


def geom_dgrad_s2(B, H, W, py, px):
    """k=4, s=2, p=1 conv with input (H, W) (even): one launch per input parity (py, px)."""
    ky0, kx0 = (py + 1) & 1, (px + 1) & 1
    oy, ox = (py + 1 - ky0) // 2, (px + 1 - kx0) // 2
    return Geom(B, H // 2, W // 2, H // 2, W // 2, 1, 1, oy, ox, -1, -1, 2, 2, 4, ky0, kx0, 2, 2, H, W, 2, 2, py, px)


def geom_pixel_shuffle2(B, H, W, i, j):
    """1x1 conv over a (H, W) map whose result lands at pixel (2y + i, 2x + j) of a (2H, 2W) map: class (i, j) of
    conv1x1 + PixelShuffle(2) (networks.py:166-175) written in place, with the output channels c * 4 + 2 i + j as its N."""
    return Geom(B, H, W, H, W, 1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 0, 1, 1, 2 * H, 2 * W, 2, 2, i, j)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ---------------------------------------------------------------------------------------------- conv
def _conv_args(x, w, geom, N, Cc, w_sn, w_sc, out, scale=None, bias=None, add1=None, add2=None, act=ACT_NONE,
               mask=None, mask_slope=0.0, scale2=None, scale_split=0, out2=None, count=True, pack=True):
    a = ConvArgs()
    a.g = geom
    a.inp, a.in_ld, a.C = x.data_ptr(), ld_of(x), Cc
    w_st = 1
    if pack and (Cc % 32 == 0) and (N % 32 == 0) and (w_sc != 1 or w.data_ptr() % 16 or w_sn % 4):
        w, w_sn, w_sc, w_st = packed_weight_view(w, N, Cc, w_sn, w_sc)
    a.w, a.w_sn, a.w_sc, a.w_st, a.N = w.data_ptr(), w_sn, w_sc, w_st, N
    a.out, a.out_ld = out.data_ptr(), ld_of(out)
    a.scale, a.bias = _ptr(scale), _ptr(bias)
    a.scale2, a.scale_split = _ptr(scale2), int(scale_split) if scale2 is not None else 0
    a.add1, a.add1_ld = _ptr(add1), (ld_of(add1) if add1 is not None else 0)
    a.add2, a.add2_ld = _ptr(add2), (ld_of(add2) if add2 is not None else 0)
    a.act = act
    a.mask, a.mask_ld, a.mask_slope = _ptr(mask), (ld_of(mask) if mask is not None else 0), mask_slope
    a.out2, a.out2_ld = _ptr(out2), (ld_of(out2) if out2 is not None else 0)
    a.ws, a.ws_bytes = None, 0
    if FLOP_COUNT is not None and count:
        _count("conv_mfma" if (Cc % 32 == 0 and N % 32 == 0) else "conv_valu", 2.0 * geom.B * geom.OH * geom.OW * N * Cc * geom.TH * geom.TW)
    return a


def conv(x, w, geom, N, Cc, w_sn, w_sc, out, **kw):
    """out = epilogue(conv(x, W-view)).  `out` is an NHWC tensor (B, OHF, OWF, >=N view).  Keywords: scale, bias, add1,
    add2, act, mask, mask_slope, scale2, scale_split, out2 (see fuses_masked_cotangent: also store the value before the
    mask factor)."""
    L = _lib.lib()
    if winograd_s2_takes(geom, N, Cc, kw) and _conv_winograd_s2([((x, w, geom, N, Cc, w_sn, w_sc, out), kw)]):
        return out
    takes = winograd_takes(geom, N, Cc, kw)
    kw.pop("wino32", None)
    if takes:
        a = _conv_args(x, w, geom, N, Cc, w_sn, w_sc, out, pack=False, **kw)
        if L.mtd_conv_winograd_ok(C.byref(a)):
            wv, px = winograd_weight_view(w, N, Cc, w_sn, w_sc, geom)
            a.w, a.w_st = wv.data_ptr(), px           # (w_st tells the library which form the weights are)
            wkey = ("wino", bytes(geom), N, Cc)
            need = _igemm_ws_cache.get(wkey)
            if need is None:
                need = L.mtd_conv_winograd_ws_bytes(C.byref(a))
                _igemm_ws_cache[wkey] = need
            if need:
                ws = workspace(need, x.device)
                a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
            if FLOP_COUNT is not None:          # executed MFMA flops: 4 (F(2x2)) or 3 (F(2x4)) instead of 9 multiplications per output pixel
                saved = 2.0 * geom.B * geom.OH * geom.OW * N * Cc * (6 if (px & 15) == 6 else 5)
                FLOP_COUNT["conv_mfma"] -= saved
                FLOP_COUNT["conv_winograd_saved"] = FLOP_COUNT.get("conv_winograd_saved", 0.0) + saved
            check(L.mtd_conv_winograd(C.byref(a), stream_ptr()), "mtd_conv_winograd")
            return out
        if FLOP_COUNT is not None:
            FLOP_COUNT["conv_mfma"] -= 2.0 * geom.B * geom.OH * geom.OW * N * Cc * 9
            FLOP_COUNT["launches"] -= 1
    a = _conv_args(x, w, geom, N, Cc, w_sn, w_sc, out, **kw)
    if (Cc % 32 == 0) and (N % 32 == 0):
        wkey = (bytes(geom), N, Cc)
        need = _igemm_ws_cache.get(wkey)
        if need is None:
            need = L.mtd_conv_igemm_ws_bytes(C.byref(a))
            _igemm_ws_cache[wkey] = need
        if need:
            ws = workspace(need, x.device)
            a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
            if SPLITK_FIN:
                a.tile_ctr, a.tile_ctr_len = tile_counters(x.device).data_ptr(), TILE_CTRS
        if CALL_LOG is not None:
            CALL_LOG.append(("igemm", bytes(a)))
        check(L.mtd_conv_igemm(C.byref(a), stream_ptr()), "mtd_conv_igemm")
    else:
        if CALL_LOG is not None:
            CALL_LOG.append(("direct", bytes(a)))
        check(L.mtd_conv_direct(C.byref(a), stream_ptr()), "mtd_conv_direct")
    return out


PAIR_CONV = _options.lab("MTD_NO_PAIR_CONV", "0") != "1"
PAIR_MAX_PIXELS = int(_options.lab("MTD_PAIR_MAX_PIXELS", "0"))      # lab: pairs only on maps of at most this many pixels per launch (0: every pair)


def conv_pair(call_a, call_b):
    """conv_group of two."""
    conv_group([call_a, call_b])


def conv_group(calls):
    """Two or three conv() calls of ONE shape -- the mirror layers of the discriminator's two decoders, or the data gradients of one
    layer in backward passes that are advanced together -- as one launch where all go to the general Winograd kernels
    (mtd_conv_winograd_group); single launches otherwise.  calls: list of (args tuple, keyword dict) as conv() takes them.  Same
    results as single conv() calls up to the grouping of the K sum (the group's split of K is planned for its whole grid)."""
    (x0, w0, geom, N, Cc, _wsn, _wsc, _o0), _kw0 = calls[0]
    L = _lib.lib()
    n = len(calls)
    same = all(bytes(c[0][2]) == bytes(geom) and (c[0][3], c[0][4]) == (N, Cc) for c in calls[1:])
    if (PAIR_CONV and 2 <= n <= 3 and same and N % 64 == 0 and (PAIR_MAX_PIXELS == 0 or geom.B * geom.OH * geom.OW <= PAIR_MAX_PIXELS)
            and all(winograd_takes(geom, N, Cc, c[1]) for c in calls)):
        arr = (ConvArgs * n)(*[_conv_args(*c[0], pack=False, **c[1]) for c in calls])
        if all(L.mtd_conv_winograd_ok(C.byref(arr[i])) for i in range(n)):
            px = 0
            for i, (args, _kw) in enumerate(calls):
                wv, px = winograd_weight_view(args[1], N, Cc, args[5], args[6], geom)
                arr[i].w, arr[i].w_st = wv.data_ptr(), px
            wkey = ("wino", bytes(geom), N, Cc)
            need = _igemm_ws_cache.get(wkey)
            if need is None:
                need = L.mtd_conv_winograd_ws_bytes(C.byref(arr[0]))
                _igemm_ws_cache[wkey] = need
            if need:
                need = (need + 255) & ~255
                ws = workspace(n * need, x0.device)
                for i in range(n):
                    arr[i].ws, arr[i].ws_bytes = ws.data_ptr() + i * need, need
            if L.mtd_conv_winograd_group_ok(arr, n):
                if FLOP_COUNT is not None:      # (the _conv_args calls counted the direct form, one launch each)
                    saved = n * 2.0 * geom.B * geom.OH * geom.OW * N * Cc * (6 if (px & 15) == 6 else 5)
                    FLOP_COUNT["conv_mfma"] -= saved
                    FLOP_COUNT["conv_winograd_saved"] = FLOP_COUNT.get("conv_winograd_saved", 0.0) + saved
                    FLOP_COUNT["launches"] -= n - 1
                check(L.mtd_conv_winograd_group(arr, n, stream_ptr()), "mtd_conv_winograd_group")
                return
        if FLOP_COUNT is not None:              # (conv() below counts again)
            FLOP_COUNT["conv_mfma"] -= n * 2.0 * geom.B * geom.OH * geom.OW * N * Cc * 9
            FLOP_COUNT["launches"] -= n
    if n == 3 and PAIR_CONV and same:           # (a group of three that does not qualify: try the first two, the third by itself)
        conv_group(calls[:2])
        conv(*calls[2][0], **calls[2][1])
        return
    for args, kw in calls:
        conv(*args, **kw)


def conv_relu_add_ok(x, w, geom, N, Cc, w_sn, w_sc, out, **kw):
    """Would conv(..., act=ACT_RELU_ADD) -- relu(conv + bias) + add1 + add2, the adds AFTER the activation -- be taken for these
    arguments (mtd_conv_relu_add_ok)?  Nothing is launched or counted."""
    if _options.lab("MTD_NO_RELU_ADD", "0") == "1" or (Cc % 32) or (N % 32):
        return False
    kw = dict(kw, act=ACT_RELU_ADD)
    if winograd_takes(geom, N, Cc, kw):        # (the 32-channel F(2x4) form carries this epilogue: conv_winograd.hip)
        a = _conv_args(x, w, geom, N, Cc, w_sn, w_sc, out, pack=False, count=False, **kw)
        if _lib.lib().mtd_conv_winograd_ok(C.byref(a)):
            return True
    a = _conv_args(x, w, geom, N, Cc, w_sn, w_sc, out, count=False, **kw)
    return bool(_lib.lib().mtd_conv_relu_add_ok(C.byref(a)))


MULTI_CONV = _options.lab("MTD_NO_MULTI_CONV", "0") != "1"


def conv_multi(calls):
    """Up to four conv() calls of one shape (same pixels, N, C, taps; each with its own geometry offsets and operands) as
    ONE grid (mtd_conv_igemm_multi): the four input-parity classes of a stride-2 data gradient.  calls: list of
    (args tuple, keyword dict) exactly as conv() takes them."""
    N, Cc = calls[0][0][3], calls[0][0][4]
    if not MULTI_CONV or len(calls) == 1 or (Cc % 32) or (N % 32):
        for args, kw in calls:
            conv(*args, **kw)
        return
    L = _lib.lib()
    if all(winograd_s2_takes(args[2], N, Cc, kw) for args, kw in calls) and _conv_winograd_s2(calls):
        return
    arr = (ConvArgs * len(calls))(*[_conv_args(*args, **kw) for args, kw in calls])
    key = ("multi", bytes(calls[0][0][2]), N, Cc, len(calls))
    need = _igemm_ws_cache.get(key)
    if need is None:
        need = L.mtd_conv_igemm_multi_ws_bytes(arr, len(calls))
        _igemm_ws_cache[key] = need
    if need:
        need = (need + 255) & ~255
        ws = workspace(need * len(calls), calls[0][0][0].device)
        for i in range(len(calls)):
            arr[i].ws, arr[i].ws_bytes = ws.data_ptr() + i * need, need
    check(L.mtd_conv_igemm_multi(arr, len(calls), stream_ptr()), "mtd_conv_igemm_multi")


FUSE_ACT_GRAD = _options.lab("MTD_NO_FUSED_ACT_GRAD", "0") != "1"


def fuses_masked_cotangent(B, H, W, Cc, N):
    """Can a 3x3 / stride-1 conv launch of this shape write both its result and result * (mask > 0) (conv(out2=...))?
    That is the halo-tile kernel's domain (conv_igemm.hip c32t_eligible + the generator-shape test of the dispatch)."""
    return FUSE_ACT_GRAD and Cc == 32 and N % 32 == 0 and H == 64 and W == 64 and B * H * W >= 32768


class DeferredWgrads:
    """Weight-gradient slab sets of one backward pass whose sums are taken together at the end: two launches
    (conv layers, spectral-mix layers) instead of one per layer.  Each layer gets a slab buffer of its own, kept for
    the life of the process under the gradient tensor's address (a generator pass holds 41 x 9.5 MB + 21 x 9.2 MB)."""

    def __init__(self):
        self.conv, self.mix, self.bufs, self.slot = [], [], [], 0


_deferred_ws = {}


def _layer_ws(nbytes, defer, device):
    # keyed by the layer's position in the pass (the schedules are fixed), so the set of buffers is bounded whatever
    # gradient tensors the caller hands in; reuse from pass to pass is ordered by the streams like workspace()'s
    key = (defer.slot, device.index if device.index is not None else _cur_device(), CAPTURE_TAG)
    defer.slot += 1
    buf = _deferred_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _deferred_ws[key] = buf
    return buf


def flush_wgrads(defer):
    """Sum every deferred slab set into its gradient tensors (on the current stream, which must be ordered after
    the kernels that wrote the slabs)."""
    L = _lib.lib()
    if defer.bufs:
        crosses_streams(*defer.bufs)
    if defer.conv:
        blocks = 0
        for d in defer.conv:
            d.first_block = blocks
            nb = L.mtd_conv_wgrad_reduce_blocks(C.byref(d))
            if nb <= 0:
                raise RuntimeError("mtd_conv_wgrad_reduce_blocks: invalid deferred weight-gradient descriptor")
            blocks += nb
        tab, host = device_table(defer.conv, defer.bufs[0].device)
        check(L.mtd_conv_wgrad_reduce_multi(tab.data_ptr(), C.cast(host, C.c_void_p), len(defer.conv), stream_ptr()),
              "mtd_conv_wgrad_reduce_multi")
    if defer.mix:
        tab, host = device_table(defer.mix, defer.bufs[0].device)
        check(L.mtd_spec_mix_wgrad_reduce_multi(tab.data_ptr(), C.cast(host, C.c_void_p), len(defer.mix), stream_ptr()),
              "mtd_spec_mix_wgrad_reduce_multi")
    defer.conv, defer.mix, defer.bufs, defer.slot = [], [], [], 0


# Lab switch, off: the block conv's weight-gradient launch carries the row transform of the same cotangent (workgroups of
# both kinds in one grid, mtd_conv_wgrad_slabs_rfft).  Correct (bit-identical, tested) but slower: the fused launch takes
# 54 us against 29.3 + 8.7 us for the two -- the row-transform waves share SIMDs with the MFMA waves on half of the CUs.
FUSE_WGRAD_ROWS = _options.lab("MTD_FUSED_WGRAD_ROWS", "0") == "1"


_half_ok_cache = {}


def wgrad_half_ok(geom, N, Cc, m_first):
    """Would wgrad(..., half=...) be taken for a layer of this shape (mtd_conv_wgrad_half_scale_ok: the register-operand small-map
    kernels, m_first a multiple of 32)?  Shapes only; nothing is launched."""
    key = (bytes(geom), N, Cc, m_first)
    ok = _half_ok_cache.get(key)
    if ok is None:
        a = WgradArgs()
        a.g = geom
        a.p = a.q = a.dw = a.half_scale = a.half_scale2 = 16            # (non-null, aligned placeholders: the query looks at shapes)
        a.p_ld, a.N, a.q_ld, a.C, a.w_sn, a.w_sc, a.m_first = N, N, Cc, Cc, Cc * geom.TH * geom.TW, geom.TH * geom.TW, int(m_first)
        ok = bool(_lib.lib().mtd_conv_wgrad_half_scale_ok(C.byref(a)))
        _half_ok_cache[key] = ok
    return ok


def wgrad(p, q, geom, N, Cc, dw, w_sn, w_sc, db=None, accumulate=False, accumulate_bias=None, defer=None, rows=None, half=None):
    """rows = (x, col_weight): also return rfft_rows(x, col_weight) -- carried by the weight-gradient launch itself when
    the slab sums are deferred (mtd_conv_wgrad_slabs_rfft), a launch of its own otherwise.
    half = (scale1, scale2, m_first): device scalars by which the cotangent of pixels [0, m_first) / the rest is multiplied as it
    is used (mtd_wgrad_args.half_scale: both halves of a paired pass, each over its own sigma, in one launch); db stays unscaled."""
    if rows is not None and not (defer is not None and DEFER_WGRADS and FUSE_WGRAD_ROWS and N % 32 == 0 and Cc % 32 == 0):
        wgrad(p, q, geom, N, Cc, dw, w_sn, w_sc, db=db, accumulate=accumulate, accumulate_bias=accumulate_bias, defer=defer)
        return rfft_rows(rows[0], rows[1])
    L = _lib.lib()
    a = WgradArgs()
    a.g = geom
    a.p, a.p_ld, a.N = p.data_ptr(), ld_of(p), N
    a.q, a.q_ld, a.C = q.data_ptr(), ld_of(q), Cc
    a.dw, a.w_sn, a.w_sc = dw.data_ptr(), w_sn, w_sc
    a.db = _ptr(db)
    if accumulate_bias is None:
        accumulate_bias = accumulate
    a.accumulate = (1 if accumulate else 0) | (2 if accumulate_bias else 0)
    a.ws, a.ws_bytes = None, 0
    if half is not None:
        a.half_scale, a.half_scale2, a.m_first = half[0].data_ptr(), half[1].data_ptr(), int(half[2])
    if FLOP_COUNT is not None:
        _count_wgrad(geom, N, Cc, L.mtd_conv_wgrad_plan_cfg(C.byref(a)))
    need = L.mtd_conv_wgrad_ws_bytes(C.byref(a))
    if need == 0:
        raise RuntimeError(f"mtd_conv_wgrad: unsupported arguments N={N} C={Cc}")
    if defer is not None and DEFER_WGRADS and N % 32 == 0 and Cc % 32 == 0:
        ws = _layer_ws(need, defer, p.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        nslab, stride = C.c_int(0), C.c_longlong(0)
        R = None
        if rows is not None:
            xr = rows[0]
            R = torch.empty((xr.shape[0], 33, 64, 64), dtype=torch.float32, device=xr.device)
            check(L.mtd_conv_wgrad_slabs_rfft(C.byref(a), C.byref(nslab), C.byref(stride), xr.data_ptr(), ld_of(xr), R.data_ptr(),
                                              xr.shape[0], int(rows[1]), stream_ptr()), "mtd_conv_wgrad_slabs_rfft")
        else:
            check(L.mtd_conv_wgrad_slabs(C.byref(a), C.byref(nslab), C.byref(stride), stream_ptr()), "mtd_conv_wgrad_slabs")
        if nslab.value > 0:
            d = _lib.WgradReduceDesc()
            d.a, d.T, d.nslab, d.slab_stride = a, geom.TH * geom.TW, nslab.value, stride.value
            d.a.p, d.a.q = None, None          # not read by the reduce; keeps the table's bytes (its cache key) the same from step to step
            defer.conv.append(d)
            defer.bufs.append(ws)
        return R
    ws = workspace(need, p.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    if CALL_LOG is not None:
        CALL_LOG.append(("wgrad", bytes(a)))
    check(L.mtd_conv_wgrad(C.byref(a), stream_ptr()), "mtd_conv_wgrad")


WGRAD_SUM = _options.lab("MTD_WGRAD_SUM", "1") == "1"      # second cotangent added inside the pair launch (0: by kernels.add)


def wgrad_pair(p, q, geom, b_first, N, Cc, dw1, dw2, w_sn, w_sc, db=None, accumulate_bias=True, p_add=None):
    """The raw weight gradients of the two image ranges [0, b_first), [b_first, B) of one batch (a paired discriminator
    pass: each range has its own spectral-norm statistics) into dw1 / dw2 and the sum of both bias gradients into db: ONE
    launch of the slab-producing kernel where the library's plan allows it (mtd_conv_wgrad_pair), two launches otherwise.
    geom: the forward geometry of the whole batch.  p_add: a second cotangent -- the gradients are those of p + p_add; added
    inside the launch where the plan can (mtd_conv_wgrad_pair_sum), by a pass of its own otherwise."""
    L = _lib.lib()
    if p_add is not None and (ld_of(p_add) != ld_of(p) or not WGRAD_SUM):
        p, p_add = add(p, p_add), None
    a = WgradArgs()
    a.g = geom
    a.p, a.p_ld, a.N = p.data_ptr(), ld_of(p), N
    a.q, a.q_ld, a.C = q.data_ptr(), ld_of(q), Cc
    a.dw, a.w_sn, a.w_sc = dw1.data_ptr(), w_sn, w_sc
    a.db = _ptr(db)
    a.accumulate = 2 if accumulate_bias else 0
    a.ws, a.ws_bytes = None, 0
    need = L.mtd_conv_wgrad_pair_ws_bytes(C.byref(a), b_first)       # 0: this layer's plan has no pair form
    if p_add is not None and (need == 0 or L.mtd_conv_wgrad_pair_ok(C.byref(a), b_first) != 2):
        p, p_add = add(p, p_add), None
        a.p = p.data_ptr()
    if need == 0:
        B = geom.B
        ga, gb = mtd_geom_with_batch(geom, b_first), mtd_geom_with_batch(geom, B - b_first)
        wgrad(p[:b_first], q[:b_first], ga, N, Cc, dw1, w_sn, w_sc, db=db, accumulate=False, accumulate_bias=accumulate_bias)
        wgrad(p[b_first:], q[b_first:], gb, N, Cc, dw2, w_sn, w_sc, db=db, accumulate=False, accumulate_bias=True)
        return
    if FLOP_COUNT is not None:
        _count_wgrad(geom, N, Cc, L.mtd_conv_wgrad_plan_cfg(C.byref(a)) if L.mtd_conv_wgrad_pair_ok(C.byref(a), b_first) == 2 else -1)      # (2: the Winograd kernels' pair form)
    ws = workspace(need, p.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    check(L.mtd_conv_wgrad_pair_sum(C.byref(a), _ptr(p_add), dw2.data_ptr(), b_first, stream_ptr()), "mtd_conv_wgrad_pair_sum")


def mtd_geom_with_batch(geom, B):
    g = type(geom)()
    C.memmove(C.byref(g), C.byref(geom), C.sizeof(g))
    g.B = B
    return g


FUSE_C32_BWD = _options.lab("MTD_NO_FUSED_C32_BWD", "0") != "1"
UNFUSE_PLAIN_C32_BWD = _options.lab("MTD_UNFUSE_PLAIN_C32_BWD", "1") == "1"


def conv_wgrad_fusable(conv_call, wgrad_call):
    """Would conv_wgrad_fused take this pair (given deferred slab sums)?  Nothing is launched or counted."""
    (p_, q_, geom, N, Cc, dw, w_sn, w_sc), wkw = wgrad_call
    if not (FUSE_C32_BWD and DEFER_WGRADS and N == 32 and Cc == 32 and not wkw.get("accumulate")):
        return False
    d = _conv_args(*conv_call[0], count=False, **{k: v for k, v in conv_call[1].items() if k != "wino32"})
    a = WgradArgs()
    a.g = geom
    a.p, a.p_ld, a.N = p_.data_ptr(), ld_of(p_), N
    a.q, a.q_ld, a.C = q_.data_ptr(), ld_of(q_), Cc
    a.dw, a.w_sn, a.w_sc = dw.data_ptr(), w_sn, w_sc
    a.db = _ptr(wkw.get("db"))
    return bool(_lib.lib().mtd_conv_c32_bwd_ok(C.byref(d), C.byref(a)))


def conv_wgrad_fused(conv_call, wgrad_call, defer, spec=None):
    """The data gradient and the weight gradient of one 32 -> 32 channel 3x3 generator layer in ONE launch
    (mtd_conv_c32_bwd: the eight waves of a workgroup split by role) when the pair is eligible and the weight-gradient
    slab sums are deferred; otherwise the two launches, the weight gradient through `side` as before.
    conv_call = (args, kw) of conv(); wgrad_call = (args, kw) of wgrad() without `defer`.  Returns True if fused.
    spec = gT (output of spec_mix_bwd): the launch closes a Res-FFT-Conv block's backward pass, its data gradient also takes
    irfft_rows(gT) (mtd_conv_c32_bwd_irfft instead of a mtd_irfft_rows launch)."""
    (p_, q_, geom, N, Cc, dw, w_sn, w_sc), wkw = wgrad_call
    if not (FUSE_C32_BWD and defer is not None and DEFER_WGRADS and N == 32 and Cc == 32 and not wkw.get("accumulate")):
        return False
    L = _lib.lib()
    d = _conv_args(*conv_call[0], **{k: v for k, v in conv_call[1].items() if k != "wino32"})
    a = WgradArgs()
    a.g = geom
    a.p, a.p_ld, a.N = p_.data_ptr(), ld_of(p_), N
    a.q, a.q_ld, a.C = q_.data_ptr(), ld_of(q_), Cc
    a.dw, a.w_sn, a.w_sc = dw.data_ptr(), w_sn, w_sc
    db = wkw.get("db")
    a.db = _ptr(db)
    a.accumulate = 0
    a.ws, a.ws_bytes = None, 0
    # Round 5: a layer WITHOUT a spectral tail whose weight gradient the library plans on the Winograd 32 x 32 kernel (plan 19,
    # csrc/conv_wgrad_wino32.h: 26 instead of 35 us) goes as two launches -- the halo-tile data gradient with all eight waves + that
    # kernel on the side stream: generator leg 5.33 -> 5.21 ms, step -0.04 ... -0.09 ms (profiles/r5_wgrad32_probe.txt)
    unfuse = False
    if spec is None and UNFUSE_PLAIN_C32_BWD:
        ukey = ("c32 unfuse", bytes(geom))
        unfuse = _igemm_ws_cache.get(ukey)
        if unfuse is None:
            unfuse = L.mtd_conv_wgrad_plan_cfg(C.byref(a)) == WGRAD_CFG_WINO32
            _igemm_ws_cache[ukey] = unfuse
    if unfuse or not L.mtd_conv_c32_bwd_ok(C.byref(d), C.byref(a)):
        if FLOP_COUNT is not None:
            FLOP_COUNT["conv_mfma"] -= 2.0 * geom.B * geom.OH * geom.OW * N * Cc * 9      # (_conv_args counted it; conv() will again)
            FLOP_COUNT["launches"] -= 1
        return False
    if FLOP_COUNT is not None:
        _count("wgrad_mfma", 2.0 * geom.B * geom.OH * geom.OW * N * Cc * 9)
        FLOP_COUNT["launches"] -= 1                                                          # one launch for the two
    need = max(L.mtd_conv_c32_bwd_ws_bytes(C.byref(d), C.byref(a)), L.mtd_conv_wgrad_ws_bytes(C.byref(a)))
    ws = _layer_ws(need, defer, p_.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    nslab, stride = C.c_int(0), C.c_longlong(0)
    if spec is not None:
        _count("fft", spec.shape[0] * 32 * _FFT_HALF_PLANE)
        if FLOP_COUNT is not None:
            FLOP_COUNT["launches"] -= 1
        check(L.mtd_conv_c32_bwd_irfft(C.byref(d), C.byref(a), spec.data_ptr(), C.byref(nslab), C.byref(stride), stream_ptr()),
              "mtd_conv_c32_bwd_irfft")
    else:
        check(L.mtd_conv_c32_bwd(C.byref(d), C.byref(a), C.byref(nslab), C.byref(stride), stream_ptr()), "mtd_conv_c32_bwd")
    r = _lib.WgradReduceDesc()
    r.a, r.T, r.nslab, r.slab_stride = a, 9, nslab.value, stride.value
    r.a.p, r.a.q = None, None
    defer.conv.append(r)
    defer.bufs.append(ws)
    return True


# ---------------------------------------------------------------------------------------------- spectral path
def rfft_rows(x, col_weight):
    B = x.shape[0]
    _count("fft", B * 32 * _FFT_HALF_PLANE)
    R = torch.empty((B, 33, 64, 64), dtype=torch.float32, device=x.device)
    check(_lib.lib().mtd_rfft_rows(x.data_ptr(), ld_of(x), R.data_ptr(), B, int(col_weight), stream_ptr()), "mtd_rfft_rows")
    return R


SPECMIX4 = _options.lab("MTD_SPECMIX4", "1") != "0"      # four-wave column / mix kernels (csrc/resfft4.hip); 0 = the one-wave forms


def spec_mix_fwd(R, w2t, b2, save):
    """Returns (T, S, Z): Z is the saved pre-activation -- its sign mask (uint8 tensor) with the four-wave kernels, the float
    tensor with the one-wave kernels; spec_mix_bwd takes whichever the forward produced."""
    B = R.shape[0]
    _count("spec_mix_mfma", B * 2112 * 2.0 * 64 * 64)
    _count("fft", 2 * B * 32 * _FFT_HALF_PLANE)
    T = torch.empty_like(R)
    S = torch.empty_like(R) if save else None
    L = _lib.lib()
    if SPECMIX4:
        Z = torch.empty(L.mtd_spec_mix_zmask_bytes(B), dtype=torch.uint8, device=R.device) if save else None
        check(L.mtd_spec_mix_fwd4(R.data_ptr(), w2t.data_ptr(), b2.data_ptr(), T.data_ptr(), _ptr(S), _ptr(Z), B, stream_ptr()),
              "mtd_spec_mix_fwd4")
        return T, S, Z
    Z = torch.empty_like(R) if save else None
    check(L.mtd_spec_mix_fwd(R.data_ptr(), w2t.data_ptr(), b2.data_ptr(), T.data_ptr(), _ptr(S), _ptr(Z), B, stream_ptr()),
          "mtd_spec_mix_fwd")
    return T, S, Z


def spec_mix_bwd(gR, w2, S, Z, dw2, db2, accumulate=False, defer=None):
    L = _lib.lib()
    B = gR.shape[0]
    _count("spec_mix_mfma", 2 * B * 2112 * 2.0 * 64 * 64)
    _count("fft", 2 * B * 32 * _FFT_HALF_PLANE)
    gT = torch.empty_like(gR)
    deferred = defer is not None and DEFER_WGRADS and dw2.data_ptr() % 16 == 0 and B * 17 <= 4096
    need = L.mtd_spec_mix_bwd_ws_bytes(B)
    ws = _layer_ws(need, defer, gR.device) if deferred else workspace(need, gR.device)
    if Z.dtype == torch.uint8:
        check(L.mtd_spec_mix_bwd4(gR.data_ptr(), w2.data_ptr(), S.data_ptr(), Z.data_ptr(), gT.data_ptr(), ws.data_ptr(), B, stream_ptr()),
              "mtd_spec_mix_bwd4")
    else:
        check(L.mtd_spec_mix_bwd(gR.data_ptr(), w2.data_ptr(), S.data_ptr(), Z.data_ptr(), gT.data_ptr(), ws.data_ptr(), B, stream_ptr()),
              "mtd_spec_mix_bwd")
    if deferred:
        d = _lib.MixReduceDesc()
        d.ws, d.dw2, d.db2, d.nslab, d.accumulate = ws.data_ptr(), dw2.data_ptr(), db2.data_ptr(), B * 17, 1 if accumulate else 0
        defer.mix.append(d)
        defer.bufs.append(ws)
        return gT
    check(L.mtd_spec_mix_wgrad_reduce(ws.data_ptr(), B, dw2.data_ptr(), db2.data_ptr(), 1 if accumulate else 0, stream_ptr()),
          "mtd_spec_mix_wgrad_reduce")
    return gT


def irfft_rows(T, out, add1=None, add2=None, mask=None):
    B = T.shape[0]
    _count("fft", B * 32 * _FFT_HALF_PLANE)
    check(_lib.lib().mtd_irfft_rows(T.data_ptr(), out.data_ptr(), ld_of(out), _ptr(add1), ld_of(add1) if add1 is not None else 0,
                                    _ptr(add2), ld_of(add2) if add2 is not None else 0, _ptr(mask),
                                    ld_of(mask) if mask is not None else 0, B, stream_ptr()), "mtd_irfft_rows")
    return out


CH32 = 32
BLOCK_TAIL = _options.lab("MTD_NO_BLOCK_TAIL", "0") != "1"      # conv3x3 + inverse row transform + residual in one launch


# Round 6: a Res-FFT-Conv block's 3x3 conv on the persistent F(2x4, 3x3) kernel too -- in the forward pass (BLOCK_FWD_WINO: that kernel +
# the closing row transform as a launch of its own, instead of the halo-tile kernel with the transform in its tail) and as its data
# gradient (BLOCK_BWD_WINO: that kernel, the Winograd 32 x 32 weight gradient and the closing row transform instead of the fused c32_bwd
# launch).  The persistent kernel wants the chip to itself: beside another stream's kernels its workgroups wait for CUs, and the forms
# with side streams (1) win 0.11 ms in the generator leg and LOSE 0.2-0.45 ms in the full step; on ONE stream (forward 2, backward 3:
# the weight gradient on the same stream too) the leg gains 0.12 ms (5.09 -> 4.98 ms) and the step 0.01-0.08 ms (four A/B pairs).
# 0: the fused launches of rounds 3-5.
BLOCK_FWD_WINO = int(_options.lab("MTD_BLOCK_FWD_WINO", "2"))      # 1: the conv on a side stream beside the spectral branch; 2: one stream
BLOCK_BWD_WINO = int(_options.lab("MTD_BLOCK_BWD_WINO", "3"))      # 1: spectral chain on a side stream; 2: main stream, weight gradient beside it; 3: one stream


def block_tail_ok(x, w, geom, img, bias):
    """Does the layer qualify for mtd_resfft_block_tail (halo-tile kernel: 32 -> 32 channels, 64 x 64 maps, >= 32768 pixels)?"""
    a = _conv_args(x, w, geom, CH32, CH32, CH32 * 9, 9, img, bias=bias, act=ACT_RELU, out2=img, count=False)
    return bool(_lib.lib().mtd_resfft_block_tail_ok(C.byref(a)))


def block_tail(x, w, geom, T, img, out, bias=None, act=ACT_RELU):
    """img = act(conv3x3(x) + bias); out = x + img + irfft_rows(T) in one launch (mtd_resfft_block_tail)."""
    L = _lib.lib()
    a = _conv_args(x, w, geom, CH32, CH32, CH32 * 9, 9, out, bias=bias, act=act, out2=img)
    _count("fft", x.shape[0] * 32 * _FFT_HALF_PLANE)
    if FLOP_COUNT is not None:
        FLOP_COUNT["launches"] -= 1                          # one launch for the conv and the row transform
    check(L.mtd_resfft_block_tail(C.byref(a), T.data_ptr(), stream_ptr()), "mtd_resfft_block_tail")
    return out


def spectral_branch_any(x, w2t, b2, out, add1=None, add2=None):
    """out = add1 + add2 + irfft2(relu(W2 . rfft2(x) + b2)) for a square NHWC map of side 128 / 256 / 512 (forward only)."""
    L = _lib.lib()
    B, S = x.shape[0], x.shape[1]
    R = torch.empty((B, S // 2 + 1, S, 64), dtype=torch.float32, device=x.device)
    T = torch.empty_like(R)
    check(L.mtd_rfft_rows_any(x.data_ptr(), ld_of(x), R.data_ptr(), B, S, stream_ptr()), "mtd_rfft_rows_any")
    check(L.mtd_spec_mix_any(R.data_ptr(), w2t.data_ptr(), b2.data_ptr(), T.data_ptr(), B, S, stream_ptr()), "mtd_spec_mix_any")
    check(L.mtd_irfft_rows_any(T.data_ptr(), out.data_ptr(), ld_of(out), _ptr(add1), ld_of(add1) if add1 is not None else 0,
                               _ptr(add2), ld_of(add2) if add2 is not None else 0, B, S, stream_ptr()), "mtd_irfft_rows_any")
    return out


def transpose64_all(weights):
    """Transposes of many 64 x 64 matrices in ONE launch (the mix weights of all Res-FFT blocks of a forward pass), cached
    until the weights change (dropped by weights_changed() like the packed conv weights).  Returns {id(w): transposed}."""
    out, todo = {}, []
    for w in weights:
        key = (w.data_ptr(), w._version, _pack_epoch)
        hit = _t64_cache.get(key)
        if hit is None:
            dst = _view_buffer((w.data_ptr(), "t64"), (64, 64), w.device)
            d = _lib.PtrPair()
            d.src, d.dst = w.data_ptr(), dst.data_ptr()
            todo.append(d)
            hit = (dst, w)
            _remember(_t64_cache, w.data_ptr(), key, hit)
        out[id(w)] = hit[0]
    if todo:
        tab, _host = device_table(todo, weights[0].device)
        check(_lib.lib().mtd_transpose64_multi(tab.data_ptr(), len(todo), stream_ptr()), "mtd_transpose64_multi")
    return out


def transpose64(src):
    dst = torch.empty((64, 64), dtype=torch.float32, device=src.device)
    check(_lib.lib().mtd_transpose64(src.data_ptr(), dst.data_ptr(), stream_ptr()), "mtd_transpose64")
    return dst


# ---------------------------------------------------------------------------------------------- element-wise
def act_grad(g, y, slope, out=None):
    B, H, W, Cc = g.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
    check(_lib.lib().mtd_act_grad(g.data_ptr(), ld_of(g), y.data_ptr(), ld_of(y), out.data_ptr(), ld_of(out), B * H * W, Cc,
                                  float(slope), stream_ptr()), "mtd_act_grad")
    return out


def copy_channels(a, out, accumulate=False):
    B, H, W, Cc = a.shape
    check(_lib.lib().mtd_copy_channels(a.data_ptr(), ld_of(a), out.data_ptr(), ld_of(out), B * H * W, Cc, 1 if accumulate else 0,
                                       stream_ptr()), "mtd_copy_channels")
    return out


def upsample2x_fwd(x, out):
    B, H, W, Cc = x.shape
    check(_lib.lib().mtd_upsample2x_fwd(x.data_ptr(), ld_of(x), out.data_ptr(), ld_of(out), B, H, W, Cc, stream_ptr()), "mtd_upsample2x_fwd")
    return out


def upsample2x_bwd(gout, gin, mask=None, slope=0.0):
    """gin = adjoint of the bilinear x2 upsampling applied to gout; with mask: times the LeakyReLU gradient (mask > 0 ? 1 : slope)
    of the layer that produced the upsampled tensor (one launch instead of two when the tensors allow 16-byte accesses)."""
    B, H, W, Cc = gin.shape
    L = _lib.lib()
    rc = L.mtd_upsample2x_bwd_masked(gout.data_ptr(), ld_of(gout), gin.data_ptr(), ld_of(gin), _ptr(mask), ld_of(mask) if mask is not None else 0,
                                     float(slope), B, H, W, Cc, stream_ptr())
    if rc == 0:
        return gin
    if rc != -2:                                         # anything but MTD_EALIGN is an error
        check(rc, "mtd_upsample2x_bwd_masked")
    check(L.mtd_upsample2x_bwd(gout.data_ptr(), ld_of(gout), gin.data_ptr(), ld_of(gin), B, H, W, Cc, stream_ptr()), "mtd_upsample2x_bwd")
    return act_grad(gin, mask, slope) if mask is not None else gin


def pixel_shuffle2_fwd(x, out):
    B, H, W, C4 = x.shape
    check(_lib.lib().mtd_pixel_shuffle2_fwd(x.data_ptr(), ld_of(x), out.data_ptr(), ld_of(out), B, H, W, C4 // 4, stream_ptr()),
          "mtd_pixel_shuffle2_fwd")
    return out


def pixel_shuffle2_bwd(gout, gin):
    B, H, W, C4 = gin.shape
    check(_lib.lib().mtd_pixel_shuffle2_bwd(gout.data_ptr(), ld_of(gout), gin.data_ptr(), ld_of(gin), B, H, W, C4 // 4, stream_ptr()),
          "mtd_pixel_shuffle2_bwd")
    return gin


def mul(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(_lib.lib().mtd_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), stream_ptr()), "mtd_mul")
    return out


def add(a, b):
    """a + b for two contiguous tensors of the same shape (new tensor)."""
    if a.shape != b.shape or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("add: expected two contiguous tensors of the same shape")
    out = torch.empty_like(a)
    check(_lib.lib().mtd_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), stream_ptr()), "mtd_add")
    return out


def dropout_mask(r, p, out):
    """out = (r >= p) / (1 - p): nn.Dropout(p)'s multiplier from uniform draws r (mtd_dropout_mask)."""
    check(_lib.lib().mtd_dropout_mask(r.data_ptr(), float(p), 1.0 / (1.0 - float(p)), out.data_ptr(), r.numel(), stream_ptr()), "mtd_dropout_mask")
    return out


def scale_by(a, s, out=None):
    """out = a * s[0] (s: a one-element device tensor)."""
    if out is None:
        out = torch.empty_like(a)
    check(_lib.lib().mtd_scale_by(a.data_ptr(), s.data_ptr(), out.data_ptr(), a.numel(), stream_ptr()), "mtd_scale_by")
    return out


def scalar_sums(parts, device):
    """parts: list of (a, b) -- b may be None -- of small contiguous fp32 tensors; returns the vector of sum(a) + sum(b)
    (one launch: the stacked task losses, the logged values of an iteration)."""
    structs = []
    for a, b in parts:
        d = _lib.SumDesc()
        d.a, d.na = a.data_ptr(), a.numel()
        d.b, d.nb = (b.data_ptr(), b.numel()) if b is not None else (None, 0)
        structs.append(d)
    tab, _host = device_table(structs, device)
    out = torch.empty(len(parts), dtype=torch.float32, device=device)
    check(_lib.lib().mtd_scalar_sums(tab.data_ptr(), len(parts), out.data_ptr(), stream_ptr()), "mtd_scalar_sums")
    return out


def zero_multi(tensors):
    """Zero many (small) contiguous fp32 tensors in one launch."""
    structs = []
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("zero_multi: contiguous tensors only")
        d = _lib.ZeroDesc()
        d.p, d.n = t.data_ptr(), t.numel()
        structs.append(d)
    tab, host = device_table(structs, tensors[0].device)
    check(_lib.lib().mtd_zero_multi(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), stream_ptr()), "mtd_zero_multi")


def checksum_multi(tensors):
    """Order-independent integer checksums (sum of the 32-bit patterns modulo 2^64) of contiguous fp32 tensors, one launch;
    returns an int64 device tensor with one entry per tensor."""
    structs = []
    for t in tensors:
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("checksum_multi: contiguous fp32 tensors only")
        d = _lib.ZeroDesc()
        d.p, d.n = t.data_ptr(), t.numel()
        structs.append(d)
    tab, host = device_table(structs, tensors[0].device)
    out = torch.empty(len(structs), dtype=torch.int64, device=tensors[0].device)
    check(_lib.lib().mtd_checksum_multi(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), out.data_ptr(), stream_ptr()), "mtd_checksum_multi")
    return out


# ---------------------------------------------------------------------------------------------- descriptor tables
_desc_cache = {}


class _Arena:
    """Pinned-host staging + device storage for descriptor tables.  A table is written into the pinned half and
    moved to the device half by mtd_upload (a kernel that reads the mapped pinned memory), so there is no
    hipMemcpyAsync in the step.  Bump allocation; the eager arena recycles from the start after a device
    synchronisation (every few hundred steps), the capture arena (tables referenced by a hipGraph) never does."""

    def __init__(self, device, nbytes, recycle):
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.ofs = 0
        self.recycle = recycle
        self.retired = []       # full chunks of a non-recycling arena: graphs / lists still read their tables

    def take(self, nbytes):
        n = (nbytes + 255) & ~255
        if self.ofs + n > self.host.numel():
            if not self.recycle:
                # tables of captured graphs / recorded lists are never recycled: start a new chunk; the full one is kept
                # (self.retired) for the life of the process -- a hipGraph holds only addresses.  Not while a capture is running.
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("descriptor arena of the captured graph exhausted")
                size = max(self.host.numel(), 2 * n)
                self.retired.append((self.host, self.dev))
                self.host = torch.empty(size, dtype=torch.uint8).pin_memory()
                self.dev = torch.empty(size, dtype=torch.uint8, device=self.dev.device)
                self.ofs = 0
            else:
                torch.cuda.synchronize()             # every kernel that reads an old table has finished
                for k in [k for k in _desc_cache if not k[1]]:      # (static tables live in the other arena and stay)
                    del _desc_cache[k]
                self.ofs = 0
        o = self.ofs
        self.ofs += n
        return self.host[o:o + n], self.dev[o:o + n]


_arenas = {}


def _static_tables():
    """Tables made now must outlive the step: a hipGraph capture or a LaunchList recording is in progress."""
    return RECORDING is not None or torch.cuda.is_current_stream_capturing()


def arena(device):
    capturing = _static_tables()
    key = (device.index if device.index is not None else torch.cuda.current_device(), capturing)
    a = _arenas.get(key)
    if a is None:
        if capturing:
            raise RuntimeError("the capture arena must exist before a hipGraph capture starts (call kernels.prepare_capture)")
        a = _Arena(device, 16 << 20, True)
        _arenas[key] = a
    return a


def prepare_capture(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), True)
    if key not in _arenas:
        _arenas[key] = _Arena(device, 8 << 20, False)


def device_table(structs, device):
    """Array of ctypes structs -> device memory, once per distinct content (pointers included); returns
    (device tensor, host ctypes array).  Keeps the host array alive for the C call."""
    n = len(structs)
    arr = (type(structs[0]) * n)(*structs)
    raw = bytes(arr)
    # (the stream is part of the key: a table uploaded on one stream must not be read by a kernel on another stream that
    # is not ordered after the upload)
    key = (device.index, _static_tables(), CAPTURE_TAG, _raw_stream(device.index if device.index is not None else _cur_device()), raw)
    hit = _desc_cache.get(key)
    STATS["table_hit" if hit is not None else "table_miss"] += 1
    if hit is None:
        host, dev = arena(device).take(len(raw))
        host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        # (a table's contents never change and the static arena is never recycled: the upload runs NOW and is not part of a
        # recorded list -- 27 launches per replayed iteration otherwise)
        rec_, _lib.RECORDER = _lib.RECORDER, None
        try:
            check(_lib.lib().mtd_upload(host.data_ptr(), dev.data_ptr(), host.numel(), stream_ptr()), "mtd_upload")
        finally:
            _lib.RECORDER = rec_
        hit = (dev, arr, host)
        _desc_cache[key] = hit
        if RECORDING is not None:
            RECORDING.keep.append(hit)        # the host array is an argument of the recorded call
    return hit[0], hit[1]


class HostScalars:
    """A few 4-byte values that the host rewrites before every step (the PCGrad shuffle order; the AdamW
    bias-correction terms of a captured step) and a kernel reads at execution time.  They live in pinned host
    memory, which the GPU addresses directly (zero-copy over PCIe: a handful of bytes read by one or a few
    workgroups) -- no H2D copy, which on ROCm costs a ~0.3 ms bubble in the stream per copy.  The pinned source
    must not be rewritten before the reading kernel has run: eager steps rotate through a small ring of slots
    (each guarded by an event recorded after the consumer was enqueued), a captured hipGraph is pinned to ONE
    slot and its owner (train_step.GraphedTrainStep) waits for the previous replay before set_inplace()."""

    RING = 4

    def __init__(self, device, count, dtype):
        self.slots = [torch.zeros(count, dtype=dtype).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.cur = 0

    def set(self, values):
        if not torch.cuda.is_current_stream_capturing():
            self.cur = (self.cur + 1) % self.RING
            ev = self.events[self.cur]
            if ev is not None:
                ev.synchronize()
        self.set_inplace(values)

    def set_inplace(self, values):
        host = self.slots[self.cur]
        for i, v in enumerate(values):
            host[i] = v

    def device_ptr(self):
        """Pointer a kernel may dereference (pinned memory is mapped into the device address space)."""
        STATS["zero_copy_reads"] += 1
        if RECORDING is not None and all(self is not sl for sl in RECORDING.slots):
            RECORDING.slots.append(self)             # a recorded launch reads this slot: LaunchList patches its address per replay
        return self.slots[self.cur].data_ptr()

    def next_for_replay(self, values):
        """A launch-list replay: move to the next slot of the ring (waiting, if it is still pending, for the launch that read
        it RING replays ago), write the values, return the address the recorded launch must be given this time."""
        self.cur = (self.cur + 1) % self.RING
        ev = self.events[self.cur]
        if ev is not None:
            ev.synchronize()
            self.events[self.cur] = None
        self.set_inplace(values)
        return self.slots[self.cur].data_ptr()

    def consumed(self):
        """Call after the kernel that reads the slot has been enqueued on the current stream."""
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.cur] = ev

    @property
    def host(self):
        return self.slots[self.cur]


# ---------------------------------------------------------------------------------------------- losses
def loss_terms(terms, device):
    """terms: list of _lib.LossTerm.  Returns a float tensor [len(terms)] of scale * sum(term)."""
    L = _lib.lib()
    tab, _host = device_table(terms, device)
    out = torch.empty(len(terms), dtype=torch.float32, device=device)
    ws = workspace(L.mtd_loss_terms_ws_bytes(len(terms)), device)
    check(L.mtd_loss_terms(tab.data_ptr(), len(terms), out.data_ptr(), ws.data_ptr(), stream_ptr()), "mtd_loss_terms")
    return out


def loss_term_grads(terms, device):
    tab, _host = device_table(terms, device)
    check(_lib.lib().mtd_loss_term_grads(tab.data_ptr(), len(terms), stream_ptr()), "mtd_loss_term_grads")


def make_term(kind, a, b=None, tconst=0.0, mx=None, my=None, scale=1.0, eps=0.0, grad_out=None, coef=0.0, accumulate=False):
    t = _lib.LossTerm()
    t.kind = kind
    t.a, t.b, t.tconst = a.data_ptr(), (b.data_ptr() if b is not None else None), float(tconst)
    t.mx, t.my = (mx.data_ptr() if mx is not None else None), (my.data_ptr() if my is not None else None)
    t.n, t.scale, t.eps = a.numel(), float(scale), float(eps)
    t.grad_out = grad_out.data_ptr() if grad_out is not None else None
    t.coef, t.accumulate = float(coef), 1 if accumulate else 0
    return t


def clip01(x):
    out = torch.empty_like(x)
    check(_lib.lib().mtd_clip01(x.data_ptr(), out.data_ptr(), x.numel(), stream_ptr()), "mtd_clip01")
    return out


def clip01_bwd(g, x):
    out = torch.empty_like(x)
    check(_lib.lib().mtd_clip01_bwd(g.data_ptr(), x.data_ptr(), out.data_ptr(), x.numel(), stream_ptr()), "mtd_clip01_bwd")
    return out


def edge_loss(a, b, scale, eps=1e-3, grad_out=None, coef=0.0, accumulate=False):
    """a, b: (B,64,64,1) contiguous.  Returns a 1-element tensor scale * sum sqrt(lap(a-b)^2+eps^2)."""
    L = _lib.lib()
    B = a.shape[0]
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = workspace(L.mtd_edge_loss_ws_bytes(B), a.device)
    check(L.mtd_edge_loss(a.data_ptr(), b.data_ptr(), B, float(scale), float(eps), out.data_ptr(),
                          grad_out.data_ptr() if grad_out is not None else None, float(coef), 1 if accumulate else 0,
                          ws.data_ptr(), stream_ptr()), "mtd_edge_loss")
    return out


# ---------------------------------------------------------------------------------------------- PCGrad / AdamW
def pcgrad_gram(vecs):
    L = _lib.lib()
    T, n = len(vecs), vecs[0].numel()
    gram = torch.empty(T * T, dtype=torch.float64, device=vecs[0].device)
    ws = workspace(L.mtd_pcgrad_ws_bytes(n, T), vecs[0].device)
    ptrs = [v.data_ptr() for v in vecs] + [None] * (4 - T)
    check(L.mtd_pcgrad_gram(ptrs[0], ptrs[1], ptrs[2], ptrs[3], T, n, gram.data_ptr(), ws.data_ptr(), stream_ptr()), "mtd_pcgrad_gram")
    return gram


def pcgrad_combine(vecs, gram, orders_dev, merged):
    """orders_dev: int32 device tensor, or an integer pointer the GPU can dereference (HostScalars.device_ptr())."""
    L = _lib.lib()
    optr = orders_dev if isinstance(orders_dev, int) else orders_dev.data_ptr()
    T, n = len(vecs), vecs[0].numel()
    coeff = torch.empty(4, dtype=torch.float32, device=vecs[0].device)
    ptrs = [v.data_ptr() for v in vecs] + [None] * (4 - T)
    check(L.mtd_pcgrad_combine(ptrs[0], ptrs[1], ptrs[2], ptrs[3], T, n, gram.data_ptr(), optr, merged.data_ptr(),
                               coeff.data_ptr(), stream_ptr()), "mtd_pcgrad_combine")
    return coeff


def pcgrad_coeff(gram, orders_dev, T):
    """Projection replay on the Gram matrix alone: T coefficients w with sum_i pc_i = sum_k w_k g_k (device tensor)."""
    coeff = torch.empty(4, dtype=torch.float32, device=gram.device)
    optr = orders_dev if isinstance(orders_dev, int) else orders_dev.data_ptr()
    check(_lib.lib().mtd_pcgrad_coeff(gram.data_ptr(), optr, T, coeff.data_ptr(), stream_ptr()), "mtd_pcgrad_coeff")
    return coeff


def pcgrad_axpy(vecs, coeff, scale, merged):
    """merged = scale * sum_k coeff[k] * vecs[k] (flat fp32 views of equal length, 16-byte aligned)."""
    T, n = len(vecs), vecs[0].numel()
    ptrs = [v.data_ptr() for v in vecs] + [None] * (4 - T)
    check(_lib.lib().mtd_pcgrad_axpy(ptrs[0], ptrs[1], ptrs[2], ptrs[3], T, n, coeff.data_ptr(), float(scale), merged.data_ptr(), stream_ptr()),
          "mtd_pcgrad_axpy")
    return merged


def adamw_multi_dyn(params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, dyn_ptr):
    L = _lib.lib()
    structs = []
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        t = _lib.AdamwTensor()
        t.p, t.g, t.m, t.v, t.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        structs.append(t)
    tab, host = device_table(structs, params[0].device)
    check(L.mtd_adamw_multi_dyn(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), float(beta1), float(beta2), float(eps),
                                C.c_void_p(dyn_ptr), stream_ptr()), "mtd_adamw_multi_dyn")


def adamw_multi_pre(params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, scalars):
    """scalars = (1 - lr*wd, lr / bias_correction1, 1 / sqrt(bias_correction2)) as kernel arguments."""
    L = _lib.lib()
    structs = []
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        t = _lib.AdamwTensor()
        t.p, t.g, t.m, t.v, t.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        structs.append(t)
    tab, host = device_table(structs, params[0].device)
    check(L.mtd_adamw_multi_pre(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), float(beta1), float(beta2), float(eps),
                                float(scalars[0]), float(scalars[1]), float(scalars[2]), stream_ptr()), "mtd_adamw_multi_pre")


def adamw_multi(params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, wd):
    L = _lib.lib()
    structs = []
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        t = _lib.AdamwTensor()
        t.p, t.g, t.m, t.v, t.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        structs.append(t)
    tab, host = device_table(structs, params[0].device)
    check(L.mtd_adamw_multi(tab.data_ptr(), C.cast(host, C.c_void_p), len(structs), float(lr), float(beta1), float(beta2), float(eps),
                            float(wd), int(step), stream_ptr()), "mtd_adamw_multi")


# ---------------------------------------------------------------------------------------------- side stream
SIDE_STREAMS = _options.lab("MTD_NO_SIDE_STREAMS", "0") != "1"


def set_concurrency(on):
    """Side streams on / off at run time (bench.py times single kernels with everything in one stream: under concurrency a
    launch's duration includes the time it shares the chip with other kernels).  Synchronises."""
    global SIDE_STREAMS
    torch.cuda.synchronize()
    on = bool(on) and _options.lab("MTD_NO_SIDE_STREAMS", "0") != "1"
    SIDE_STREAMS = on
    for s in _side.values():
        s.enabled = on


def _order(later, earlier):
    """Everything enqueued on `later` from now on runs after everything enqueued on `earlier` so far."""
    ev = torch.cuda.Event()
    ev.record(earlier)
    later.wait_event(ev)
    if RECORDING is not None:
        RECORDING.ops.append((ev.record, (earlier,)))
        RECORDING.ops.append((later.wait_event, (ev,)))


order_streams = _order


def rec(fn, *args):
    """A torch operation inside a recordable step (the few that are not library launches: the uniform draws of the dropout
    masks, RCCL collectives, a test hook's fill): runs now and, when a LaunchList is being recorded, on every replay.  It has
    to write into tensors that exist already (in place / out=) and read only tensors that live as long as the list."""
    fn(*args)
    if RECORDING is not None:
        RECORDING.add_call(fn, args)


class _OnStream:
    """A recorded torch operation that ran on a stream other than the list's main stream: re-issued under that stream."""

    def __init__(self, stream, fn):
        self.stream, self.fn = stream, fn

    def __call__(self, *args):
        with torch.cuda.stream(self.stream):
            self.fn(*args)


class LaunchList:
    """One step of a static-shape schedule as a flat list of (C entry point, arguments) and stream-order operations,
    recorded while the step runs eagerly and re-issued by replay() with none of the Python around the calls (tensor
    allocation, geometry and plan look-ups: 17 us per launch, which makes eager generator steps host-bound).  Unlike a
    captured hipGraph -- whose multi-stream sections ROCm 7.2 serialises -- the replay keeps the side streams, so the
    spectral branch, the weight gradients and the data-gradient chain of a Res-FFT block overlap on the chip.
    Every tensor allocated while recording is kept (no address is reused inside the step, so the stream-order
    operations of the recording are the only ordering the replay needs); scratch and descriptor tables come from the
    capture-side pools (CAPTURE_TAG, the non-recycling arena).  Inputs are the tensors the step read when it was
    recorded: refresh them in place."""

    def __init__(self):
        self.ops, self.keep = [], []
        self.main = None        # the stream that was current while recording: replay() must run under it
        self.slots = []         # HostScalars that recorded launches read (the PCGrad order, AdamW's step scalars)
        self.sites = {}         # id(slot) -> [(op index, argument index)] of the launches that take its address
        self.result = None      # what the recorded function returned (set once it has run to completion)

    def add_call(self, fn, args):
        cur = torch.cuda.current_stream()
        self.ops.append((fn if cur == self.main else _OnStream(cur, fn), args))

    def record(self, fn, device, repack=True):
        """repack: drop every derived weight view first, so that the list packs the views it reads (a list that is replayed
        while something ELSE updates the weights).  A list that contains the optimizer steps itself (train_step.
        RecordedTrainStep) keeps the views: its own pack launches rewrite them in place (_view_buffer)."""
        global RECORDING, CAPTURE_TAG
        if RECORDING is not None or torch.cuda.is_current_stream_capturing():
            raise RuntimeError("LaunchList.record: a recording or a hipGraph capture is already in progress")
        prepare_capture(device)
        if repack:
            weights_changed()             # packed / transposed weight views are produced inside the list
        torch.cuda.synchronize()
        self.main = torch.cuda.current_stream()
        real_empty, real_empty_like = torch.empty, torch.empty_like

        # (the storages are kept, not the tensors: autograd takes a gradient tensor over as .grad without a copy only
        # while nothing else refers to the tensor object)
        def empty(*a, **k):
            t = real_empty(*a, **k)
            self.keep.append(t.untyped_storage())
            return t

        def empty_like(*a, **k):
            t = real_empty_like(*a, **k)
            self.keep.append(t.untyped_storage())
            return t
        CAPTURE_TAG += 1
        RECORDING = self
        _lib.RECORDER = self.ops
        torch.empty, torch.empty_like = empty, empty_like
        try:
            out = fn()
        finally:
            torch.empty, torch.empty_like = real_empty, real_empty_like
            _lib.RECORDER = None
            RECORDING = None
        torch.cuda.synchronize()
        self.result = out             # fn ran to completion; what follows only inspects the list (RecordedTrainStep tells the two apart)
        self._find_slot_sites()
        return out

    def _find_slot_sites(self):
        """Which recorded launches take the address of a HostScalars slot (as a c_void_p or an integer argument)?"""
        self.sites = {}
        for sl in self.slots:
            ptr = sl.slots[sl.cur].data_ptr()
            found = []
            for i, (_f, args) in enumerate(self.ops):
                for j, a in enumerate(args):
                    if a.__class__ is C.c_void_p:
                        a = a.value
                    if a.__class__ is int and a == ptr:
                        found.append((i, j))
            if not found:
                raise RuntimeError("LaunchList: a pinned scalar slot was read while recording but no recorded launch takes its address")
            self.sites[id(sl)] = found

    def set_slot(self, slot, values):
        """Before a replay: new contents for a pinned slot that recorded launches read.  The slot's ring advances (the launch
        of an earlier replay may not have run yet) and the recorded launches are re-pointed."""
        ptr = slot.next_for_replay(values)
        for i, j in self.sites[id(slot)]:
            f, args = self.ops[i]
            args = list(args)
            args[j] = C.c_void_p(ptr) if args[j].__class__ is C.c_void_p else ptr
            self.ops[i] = (f, tuple(args))

    def replay(self):
        if torch.cuda.current_stream() != self.main:
            raise RuntimeError("LaunchList.replay: the current stream is not the one the list was recorded under")
        for f, args in self.ops:
            rc = f(*args)
            if rc.__class__ is int and rc:
                raise RuntimeError(f"LaunchList.replay: {getattr(f, '__name__', f)} failed with code {rc}")
        if self.slots:
            ev = torch.cuda.Event()
            ev.record()                   # every reader of the pinned slots has been enqueued before this point
            for sl in self.slots:
                sl.events[sl.cur] = ev


class SideStream:
    """A second HIP stream for work that is off the critical path of a backward pass (weight gradients, their
    slab reductions, the spectral-norm correction): it runs beside the data-gradient chain on the main stream
    so the matrix cores see two independent kernels.  fork() orders the side stream after everything enqueued
    on the main stream so far; join() makes the main stream wait for the side work.  Tensors handed to keep()
    stay referenced until join(), i.e. until the side kernels that read them are ordered before any reuse."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self._keep = []
        self.enabled = SIDE_STREAMS      # MTD_NO_SIDE_STREAMS=1 / set_concurrency(False): everything on one stream

    def fork(self):
        if self.enabled:
            _order(self.stream, torch.cuda.current_stream())

    def run(self, fn, *keep, fork=True):
        """fork=False: the work depends only on what this stream already holds (the spectral-norm correction of weight
        gradients the stream itself computed, a collective on them): no event hand-off from the main stream."""
        if not self.enabled:
            fn()
            return
        if fork:
            self.fork()
        with torch.cuda.stream(self.stream):
            fn()
        self._keep.extend(keep)

    def join(self):
        if self.enabled:
            _order(torch.cuda.current_stream(), self.stream)
        self._keep.clear()


_side = {}


def side_stream(device, idx=0):
    key = (device.index if device.index is not None else torch.cuda.current_device(), idx)
    s = _side.get(key)
    if s is None:
        s = SideStream(device)
        _side[key] = s
    return s


def crosses_streams(*tensors):
    """Tensors allocated while a side stream was current and then consumed on the main stream: tell the caching
    allocator so their memory is not recycled before the main-stream consumers have run."""
    cur = torch.cuda.current_stream()
    for t in tensors:
        if t is not None:
            t.record_stream(cur)

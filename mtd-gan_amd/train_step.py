"""The fused discriminator / generator steps behind MTD_GAN_Method.d_loss / .g_loss and
WeightMethods('pcgrad').backward (reference: arch/Ours/networks.py:1957-2009, module/weight_methods.py:
429-468, engine.py:33-55).

d_loss records the four discriminator passes on tapes and returns the stacked task losses; the PCGrad
step then replays the tapes once per task (un-stacked: no zero-cotangent traversals), writing
each task's shared-parameter gradient into its own flat vector and the task-specific gradients (only their
sum is needed) into one set of buffers; a Gram pass + a combine pass implement the projection.  g_loss is
one autograd node: G forward, one D pass without the restoration decoder, loss kernels; its backward
replays D for the input gradient only (the reference's D weight gradients in the G step are discarded by
the next zero_grad) and then the generator.  Nothing here synchronises with the host.
"""
from . import _options
import os
import random
import re

import torch

from . import discriminator_path as DP
from . import generator_path as GP
from . import kernels as K

NPIX = 64 * 64
# the first pass into a task vector overwrites its spectral-norm weight gradients (MTD_FIRST_WRITE=0: zero fill + accumulate)
FIRST_WRITE = _options.lab("MTD_FIRST_WRITE", "1") == "1"
POISON = _options.lab("MTD_POISON_TASK_VECTORS", "0") == "1"
_TRUNK_SN_WEIGHT = re.compile(r"(?:conv\d\d|down\d|bconv\d)\.weight_orig$")
re_dec = {"s": re.compile(r"s_dconv\d\d\.weight_orig$"), "r": re.compile(r"r_dconv\d\d\.weight_orig$")}
# data parallelism: ship finished gradient slices of the last task pass while it runs (MTD_DP_EARLY_SHIP=0: after the pass)
EARLY_SHIP = _options.lab("MTD_DP_EARLY_SHIP", "1") == "1"


def _nhwc1(t):
    return t.contiguous().reshape(t.shape[0], 64, 64, 1)


def _dp_avg(dp, flat, after=()):
    """dp.all_reduce_avg(flat, after=streams) -- the hook contract since round 4 (parallel.DataParallelSync): the collective
    waits for `after`, the main stream does not.  A hook object with the older one-argument form still works: the streams in
    `after` are joined into the current stream first, which is what that form assumed."""
    takes = getattr(dp, "_mtd_takes_after", None)
    if takes is None:
        import inspect
        try:
            takes = "after" in inspect.signature(dp.all_reduce_avg).parameters
        except (TypeError, ValueError):
            takes = False
        try:
            dp._mtd_takes_after = takes
        except AttributeError:
            pass
    if takes:
        return dp.all_reduce_avg(flat, after=after)
    for st in after:
        K.order_streams(torch.cuda.current_stream(), st)
    return dp.all_reduce_avg(flat)


# ================================================================================================ D step
class DStepTape:
    def __init__(self, method, P, passes, outs, x, y, fake):
        self.method, self.P, self.passes, self.outs = method, P, passes, outs
        self.x, self.y, self.fake = x, y, fake
        self.consumed = False

    def run_pcgrad(self, shared_params, task_specific_params, reduction="sum", dp=None):
        """Per-task backward + PCGrad projection.  Writes .grad of every listed parameter.
        dp: optional data-parallel hook with .all_reduce_avg(flat_tensor) (see parallel.py)."""
        D = self.method.Discriminator
        dev = self.x.device
        B = self.x.shape[0]
        n = B * NPIX
        by_id = {id(p): nme for nme, p in D.named_parameters()}
        sh_names = [by_id[id(p)] for p in shared_params]
        ts_names = [by_id[id(p)] for p in (task_specific_params or [])]
        sizes = [D.get_parameter(nme).numel() for nme in sh_names]
        total = sum(sizes)
        S = torch.empty((4, total), dtype=torch.float32, device=dev)       # 3 task vectors + merged (written whole)
        # The trunk's spectral-norm weights are 99.9 % of a task vector, and every backward pass writes all of them: the FIRST
        # pass into a vector overwrites (disc_backward(overwrite=...)) instead of adding to zeros -- no 343 MB fill per step and
        # no read of it by the correction kernel.  Only the rest (the biases, which the weight-gradient kernels accumulate)
        # is zeroed.
        first_write = frozenset() if not FIRST_WRITE else frozenset(
            nme[:-len(".weight_orig")] for nme in sh_names if _TRUNK_SN_WEIGHT.match(nme))
        if POISON:
            K.rec(S[:3].fill_, float("nan"))   # (test hook: an element no pass writes stays NaN)
        # task-specific gradients: views of one flat buffer, so that N > 1 averages them with a single collective
        ts_sizes = [D.get_parameter(nme).numel() for nme in ts_names]
        # (each view starts on a 16-byte boundary: the spectral-norm correction then moves float4s)
        TSflat = torch.empty(max(sum((sz + 3) // 4 * 4 for sz in ts_sizes), 4), dtype=torch.float32, device=dev)
        # ... and the same for the decoders' spectral-norm weights (91 % of the task-specific bucket): the SEG decoder's are first
        # written by the pass over tape 3+4, the REC decoder's by the last pass over tape 1+2 (the others hand their decoder
        # cotangents over, see below)
        dec_first = {"s": frozenset(), "r": frozenset()}
        if first_write:
            for pre in "sr":
                dec_first[pre] = frozenset(nme[:-len(".weight_orig")] for nme in ts_names if re_dec[pre].match(nme))
        if POISON:
            K.rec(TSflat.fill_, float("nan"))
        TSbuf, tofs = {}, 0
        for nme, sz in zip(ts_names, ts_sizes):
            TSbuf[nme] = TSflat[tofs:tofs + sz].view_as(D.get_parameter(nme))
            tofs += (sz + 3) // 4 * 4
        sinks = []
        for i in range(3):
            t, ofs = dict(TSbuf), 0
            for nme, sz in zip(sh_names, sizes):
                t[nme] = S[i, ofs:ofs + sz]
                ofs += sz
            sinks.append(DP.GradSink(t))
        rest = [sk.t[nme] for sk in sinks for nme in sh_names if not (nme.endswith(".weight_orig") and nme[:-len(".weight_orig")] in first_write)]
        rest += [TSbuf[nme] for nme in ts_names if not (nme.endswith(".weight_orig") and nme[:-len(".weight_orig")] in (dec_first["s"] | dec_first["r"]))]
        if first_write and rest:
            K.zero_multi(rest)                 # ~100 small tensors, one launch
        elif not first_write:
            K.rec(S[:3].zero_)
            K.rec(TSflat.zero_)
        # ---- the projection order of this step: Python's `random`, as the reference draws it.  Under data parallelism
        # it is a collective decision (rank 0's draw, broadcast on the RCCL stream under the backward passes): ranks with
        # different `random` states would otherwise project the same averaged gradients differently and drift apart.
        orders = next_orders(3)
        slot = orders_slot(dev)
        slot.set([j for o in orders for j in o] + [0] * 7)
        orders_dev = None
        if dp is not None:
            orders_dev = dp.broadcast_orders(orders, dev, upload=lambda flat: _upload_slot(slot, dev))
        (re, rd, rr), (fe, fd, fr), (rre, rrd), (rfe, rfd), r12 = self.outs
        t12, t34 = self.passes
        rt, P = D._rt, self.P
        x, y, fake = self.x, self.y, self.fake
        # ---- all output cotangents of the three tasks in one launch.  Each is allocated for the pair of passes it belongs
        # to (2B images); the real / fake halves are views, so one backward serves both passes.
        E = lambda: torch.empty((2 * B, 1, 1, 1), dtype=torch.float32, device=dev)
        Mp = lambda: torch.empty((2 * B, 64, 64, 1), dtype=torch.float32, device=dev)
        G2 = {k: (E() if k.endswith("e") else Mp()) for k in ("d_e", "c_e", "c3_e", "d_d", "r_r", "c_d", "c3_d")}
        h0 = lambda k: G2[k][:B]
        h1 = lambda k: G2[k][B:]
        g = {"d_re": h0("d_e"), "d_fe": h1("d_e"), "c_re": h0("c_e"), "c_fe": h1("c_e"), "c_rre": h0("c3_e"), "c_rfe": h1("c3_e"),
             "d_rd": h0("d_d"), "d_fd": h1("d_d"), "r_rr": h0("r_r"), "r_fr": h1("r_r"), "c_rd": h0("c_d"), "c_fd": h1("c_d"),
             "c_rrd": h0("c3_d"), "c_rfd": h1("c3_d")}
        T = K.make_term
        K.loss_term_grads([
            T(0, re, tconst=1.0, grad_out=g["d_re"], coef=1.0 / B), T(0, fe, tconst=0.0, grad_out=g["d_fe"], coef=1.0 / B),
            T(0, rd, tconst=1.0, mx=x, my=y, grad_out=g["d_rd"], coef=1.0 / n), T(0, fd, tconst=0.0, mx=x, my=y, grad_out=g["d_fd"], coef=1.0 / n),
            T(1, rr, y, grad_out=g["r_rr"], coef=1.0 / n), T(1, fr, fake, grad_out=g["r_fr"], coef=1.0 / n),
            T(0, re, rre, grad_out=g["c_re"], coef=1.0 / B), T(0, rre, re, grad_out=g["c_rre"], coef=1.0 / B),
            T(0, rd, rrd, grad_out=g["c_rd"], coef=1.0 / n), T(0, rrd, rd, grad_out=g["c_rrd"], coef=1.0 / n),
            T(0, fe, rfe, grad_out=g["c_fe"], coef=1.0 / B), T(0, rfe, fe, grad_out=g["c_rfe"], coef=1.0 / B),
            T(0, fd, rfd, grad_out=g["c_fd"], coef=1.0 / n), T(0, rfd, fd, grad_out=g["c_rfd"], coef=1.0 / n)], dev)
        sink_c = sinks[2]       # the passes run one after the other on one stream pair and all add into the one task-specific bucket

        # The decoders' gradients are sums over the tasks that reach them through passes 1+2 (SEG decoder: adversarial +
        # consistency, REC decoder: restoration + consistency) and a weight gradient is linear in its cotangent: the
        # adversarial and the restoration pass hand their decoder cotangents over and the consistency pass, the last one over
        # that tape, computes every decoder weight gradient once from the sums -- two decoder weight-gradient sweeps less.
        exp_s, exp_r = {}, {}

        def adversarial():              # image-level + pixel-level heads on passes 1 and 2
            DP.disc_backward(rt, P, t12, G2["d_e"], G2["d_d"], None, sinks[0], False, dec_export=exp_s, overwrite=first_write)

        def restoration():
            DP.disc_backward(rt, P, t12, None, None, G2["r_r"], sinks[1], False, dec_export=exp_r, overwrite=first_write)

        def consistency34():            # through D(rec.clip), passes 3 and 4 ...
            return DP.disc_backward(rt, P, t34, G2["c3_e"], G2["c3_d"], None, sink_c, True, overwrite=first_write | dec_first["s"])

        def consistency12(gin34, flush=None):         # ... and back into the restoration decoder of passes 1 and 2
            DP.disc_backward(rt, P, t12, G2["c_e"], G2["c_d"], K.clip01_bwd(gin34, r12), sink_c, False,
                             dec_import={**exp_s, **exp_r}, flush=flush, overwrite=dec_first["r"])

        # Data parallelism: the LAST pass has no later backward to hide its collectives under, so it ships what is final while
        # it runs -- the task-specific bucket (158 MB) once the decoders and heads are done (before the whole trunk), the tail
        # of the task vector (bottleneck + trunk levels 6 .. FLUSH_LEVEL: 91 % of 114 MB) once those levels are, under the
        # large-map levels that remain.  Only the head of the vector (10 MB) is reduced after the pass.
        tail_ofs, shipped = self._low_tail(sh_names, sizes), {"ts": False, "tail": False}

        def ship(stage):
            side = K.side_stream(dev)          # the weight gradients and their spectral-norm corrections are on this stream
            if stage == "heads" and ts_names:
                side.run(lambda: dp.all_reduce_avg(TSflat), fork=False)       # (ordered after the side stream's own work: no fork)
                shipped["ts"] = True
            elif stage == "trunk_low" and tail_ofs is not None:
                side.run(lambda: dp.all_reduce_avg(S[2, tail_ofs:]), fork=False)
                shipped["tail"] = True

        if DP.LOCKSTEP:
            # The adversarial pass (tape 1+2) and the first consistency pass (tape 3+4) are independent and of one structure -- image-level
            # head, pixel-level decoder, trunk: they are advanced together and the data gradients of the same layer go out as ONE launch
            # (discriminator_path.disc_backward_lockstep).  Every buffer two passes add into sees them in the order of the sequential
            # schedule (adversarial before consistency; the restoration pass touches none of those).
            p_adv = ((rt, P, t12, G2["d_e"], G2["d_d"], None, sinks[0], False), dict(dec_export=exp_s, overwrite=first_write))
            p_rest = ((rt, P, t12, None, None, G2["r_r"], sinks[1], False), dict(dec_export=exp_r, overwrite=first_write))
            p_c34 = ((rt, P, t34, G2["c3_e"], G2["c3_d"], None, sink_c, True), dict(overwrite=first_write | dec_first["s"]))
            if DP.LOCKSTEP == 3:      # (lab: all three; groups of three launches gave most of the pairs' gain back)
                _n0, _n1, gin34 = DP.disc_backward_lockstep([p_adv, p_rest, p_c34])
                self._sync_task(dp, S, 0)
            elif DP.LOCKSTEP == 2:
                # (lab: the restoration pass and the TRUNK of the second consistency pass too -- the restoration pass goes through its decoder
                # first, alone: the other pass sums its decoder cotangents)
                _n0, gin34 = DP.disc_backward_lockstep([p_adv, p_c34])
                self._sync_task(dp, S, 0)
                fl = ship if dp is not None and EARLY_SHIP else None
                DP.disc_backward_lockstep([p_rest, lambda: ((rt, P, t12, G2["c_e"], G2["c_d"], K.clip01_bwd(gin34, r12), sink_c, False),
                                                            dict(dec_import={**exp_s, **exp_r}, flush=fl, overwrite=dec_first["r"]))], lead=12)      # (12 = the 3x3 layers of its decoder)
                self._sync_task(dp, S, 1)
                gin34 = None
            else:
                _n0, gin34 = DP.disc_backward_lockstep([p_adv, p_c34])
                self._sync_task(dp, S, 0)
                restoration()                                                                # task 1
            if DP.LOCKSTEP != 2:
                self._sync_task(dp, S, 1)
                consistency12(gin34, ship if dp is not None and EARLY_SHIP else None)        # task 2, second half
        else:
            adversarial()                                                                    # task 0
            self._sync_task(dp, S, 0)
            restoration()                                                                    # task 1
            self._sync_task(dp, S, 1)
            consistency12(consistency34(), ship if dp is not None and EARLY_SHIP else None)  # task 2
        if shipped["tail"]:
            if dp is not None:
                _dp_avg(dp, S[2, :tail_ofs], after=self._side_of(dev))
        else:
            self._sync_task(dp, S, 2)
        K.side_stream(dev).join()              # weight gradients / spectral-norm corrections ran on the side stream
        if dp is not None:
            dp.wait()                          # the three averaged task vectors are needed by the Gram kernel
        if not shipped["ts"] and dp is not None and ts_names:
            dp.all_reduce_avg(TSflat)          # 158 MB, in flight under the Gram / combine kernels; joined below
        # ---- PCGrad: Gram of the original task gradients, replay of the projections, combine
        vecs = [S[0], S[1], S[2]]
        gram = K.pcgrad_gram(vecs)
        if orders_dev is not None:
            coeff = K.pcgrad_combine(vecs, gram, orders_dev, S[3])        # rank 0's order, broadcast at the top of the pass
        else:
            coeff = K.pcgrad_combine(vecs, gram, slot.device_ptr(), S[3])     # the order is read from the pinned slot at run time
            slot.consumed()
        merged = S[3]
        if reduction == "mean":
            merged = torch.empty_like(S[3])
            K.rec(lambda: torch.div(S[3], 3.0, out=merged))
        ofs = 0
        for p, sz in zip(shared_params, sizes):
            p.grad = merged[ofs:ofs + sz].view_as(p)
            ofs += sz
        for p, nme in zip(task_specific_params or [], ts_names):
            p.grad = TSbuf[nme]
        if dp is not None:
            dp.wait()
        # (under data parallelism `orders` is this rank's own draw; the order applied is rank 0's, in orders_dev)
        self.gram, self.coeff, self.orders, self.orders_dev, self.task_vectors = gram, coeff, orders, orders_dev, S
        self.consumed = True

    @staticmethod
    def _low_tail(names, sizes):
        """Offset in a task vector from which on every entry belongs to the bottleneck or to trunk levels >= FLUSH_LEVEL (the
        layers disc_backward has finished at its "trunk_low" flush), or None if those entries are not one contiguous tail."""
        import re
        low = []
        for nme in names:
            m = re.match(r"(?:conv|down)(\d)", nme)
            low.append(nme.startswith("bconv") or (m is not None and int(m.group(1)) >= DP.FLUSH_LEVEL))
        if True not in low:
            return None
        first = low.index(True)
        if not all(low[first:]) or first == 0:
            return None
        return sum(sizes[:first])

    @staticmethod
    def _side_of(dev):
        sd = K.side_stream(dev)
        return (sd.stream,) if sd.enabled else ()

    @classmethod
    def _sync_task(cls, dp, S, i):
        # The task vector is complete once the side-stream weight gradients (and their spectral-norm correction) are: the
        # COLLECTIVE waits for that stream.  The main stream does not -- it goes on with the next task's data-gradient chain
        # while this task's last weight gradients run (round 3 joined the side stream here: +1.2 ms per iteration at N = 1
        # with every collective live, profiles/r4_forced_dp_*.json).
        if dp is not None:
            _dp_avg(dp, S[i], after=cls._side_of(S.device))   # overlaps with the next task's backward (separate stream)


_orders_slots = {}
_pending_orders = []


def _upload_slot(slot, dev):
    """The 16 int32s of a pinned HostScalars slot as a device tensor, moved by mtd_upload (a kernel reading the mapped
    pinned memory: no hipMemcpy in the step)."""
    from . import _lib
    t = torch.empty(16, dtype=torch.int32, device=dev)
    K.check(_lib.lib().mtd_upload(slot.device_ptr(), t.data_ptr(), 64, K.stream_ptr()), "mtd_upload")
    slot.consumed()
    return t


def orders_slot(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _orders_slots.get(key)
    if s is None:
        s = K.HostScalars(dev, 16, torch.int32)
        _orders_slots[key] = s
    return s


def next_orders(T):
    """The shuffle order of this step: drawn from Python's `random` like the reference.  A hipGraph replay
    (GraphedTrainStep) pre-draws the order and refreshes the pinned slot itself."""
    if _pending_orders:
        return _pending_orders.pop(0)
    return shuffle_orders(T)


def shuffle_orders(T):
    """Index orders produced by the reference's cumulative in-place `random.shuffle(grads)` per i
    (module/weight_methods.py:452): same consumption of Python's `random` stream."""
    idx = list(range(T))
    out = []
    for _ in range(T):
        random.shuffle(idx)
        out.append(list(idx))
    return out


def d_loss(method, x, y):
    D, G = method.Discriminator, method.Generator
    if not x.is_cuda:
        raise RuntimeError("MTD_GAN_Method.d_loss: HIP path needs CUDA tensors")
    B, dev = x.shape[0], x.device
    from .arch.Ours import networks as N
    xn, yn = _nhwc1(x.float()), _nhwc1(y.float())
    # networks.py:1958 (.detach()).  The generator step that follows runs G on the same x with the same weights
    # (engine.py:38-55 updates G only after it), so this forward also records the tape that step needs: G runs once
    # per iteration (the "sufficient work" count of SURVEY 8d), bit-identical to recomputing it.
    gflat = G._flat_params()
    GPm = N._unflatten_gen(gflat, G._cfg[2])
    keep = torch.is_grad_enabled() and any(p.requires_grad for p in gflat)
    # (the generator writes its output straight into the second half of the first pair's input batch: no concatenation pass)
    pair12 = torch.empty((2 * B, 64, 64, 1), dtype=torch.float32, device=dev)
    K.copy_channels(yn, pair12[:B])
    fn, gtape = GP.generator_forward(xn, GPm, keep, out=pair12[B:])
    # (the entry holds x itself: a strong reference keeps its address from being recycled for another batch)
    method._gcache = (x, _gkey(x, gflat), fn, gtape, GPm) if keep else None
    P = D._param_dict()
    train = D.training
    # The four discriminator passes of the reference run as TWO launches sequences of batch 2B: (D(y), D(fake)) and
    # (D(clip(real_rec)), D(clip(fake_rec))).  Each pass keeps its own spectral-norm state: the four power iterations run
    # first, in the reference's order, and every conv applies the 1/sigma of the half it is computing (scale2 /
    # scale_split).  Same arithmetic per image; half the launches, twice the pixels per launch on the deep 4x4 .. 1x1 layers.
    K.prepack(DP.conv_views(P, True))
    K.prepack_winograd(DP.winograd_views(P, True))
    K.prepack_winograd_s2(DP.winograd_s2_views(P, True))
    sn = DP._sn_forward_multi(P, train, dev, 4)
    masks = D._next_masks(B, dev, 4)          # the four passes' dropout multipliers, stacked: (4B, 512) or None
    m12, m34 = (None, None) if masks is None else (masks[:2 * B], masks[2 * B:])
    (e12, d12, r12), t12 = DP.disc_forward(P, pair12, train, m12, True, True, sn=sn[0], sn2=sn[1], pair=B)
    (e34, d34, _), t34 = DP.disc_forward(P, K.clip01(r12), train, m34, False, True, sn=sn[2], sn2=sn[3], pair=B)
    re, fe, rd, fd, rr, fr = e12[:B], e12[B:], d12[:B], d12[B:], r12[:B], r12[B:]
    rre, rfe, rrd, rfd = e34[:B], e34[B:], d34[:B], d34[B:]
    n = B * NPIX
    T = K.make_term
    v = K.loss_terms([
        T(0, re, tconst=1.0, scale=1.0 / B), T(0, fe, tconst=0.0, scale=1.0 / B),
        T(0, rd, tconst=1.0, mx=xn, my=yn, scale=1.0 / n), T(0, fd, tconst=0.0, mx=xn, my=yn, scale=1.0 / n),
        T(1, rr, yn, scale=1.0 / n), T(1, fr, fn, scale=1.0 / n),
        T(0, re, rre, scale=1.0 / B), T(0, rd, rrd, scale=1.0 / n), T(0, fe, rfe, scale=1.0 / B), T(0, fd, rfd, scale=1.0 / n)], dev)
    losses = K.scalar_sums([(v[0:4], None), (v[4:6], None), (v[6:10], None)], dev)        # stack[disc, rec, consist] (networks.py:1992)
    keys = ["D/real_enc", "D/fake_enc", "D/real_dec", "D/fake_dec", "D/rec_loss_real", "D/rec_loss_fake",
            "D/consist_loss_real_enc", "D/consist_loss_real_dec", "D/consist_loss_fake_enc", "D/consist_loss_fake_dec"]
    details = {k: v[i] for i, k in enumerate(keys)}
    losses._mtd_tape = DStepTape(method, P, (t12, t34), ((re, rd, rr), (fe, fd, fr), (rre, rrd), (rfe, rfd), r12), xn, yn, fn)
    return losses, details


def _gkey(x, gparams):
    """Identity of a generator forward: the input tensor (storage, shape, version) and every parameter's storage and
    modification stamps (autograd version for in-place torch updates, _mtd_epoch for FusedAdamW's raw-pointer updates)."""
    return (x.data_ptr(), tuple(x.shape), x._version,
            tuple((p.data_ptr(), p._version, getattr(p, "_mtd_epoch", 0)) for p in gparams))


# ================================================================================================ G step
class _GStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, method, nlayers, *gparams):
        from .arch.Ours import networks as N
        D = method.Discriminator
        B, dev = x.shape[0], x.device
        xn, yn = _nhwc1(x.float()), _nhwc1(y.float())
        GPm = N._unflatten_gen(gparams, nlayers)
        need = any(ctx.needs_input_grad)
        cache, method._gcache_hit = getattr(method, "_gcache_hit", None), None
        if need and cache is not None:
            _, _, fake, gtape, GPm = cache                   # same input object, same weights: the D step's forward and its tape
        else:
            fake, gtape = GP.generator_forward(xn, GPm, need)
        P = D._param_dict()
        (ge, gd, _), dtape = DP.disc_forward(P, fake, D.training, D._next_mask(B, dev), False, need)
        n = B * NPIX
        T = K.make_term
        v = K.loss_terms([T(0, ge, tconst=1.0, scale=1.0 / B), T(0, gd, tconst=1.0, mx=xn, my=yn, scale=1.0 / n),
                          T(2, fake, yn, scale=50.0 / n, eps=method.pixel_loss.eps)], dev)
        edge = K.edge_loss(fake, yn, 50.0 / n, method.edge_loss.loss.eps)
        total = K.scalar_sums([(v, edge)], dev).reshape(())              # adv + 50 Charbonnier + 50 Edge (networks.py:2003)
        if need:
            ctx.state = (method, GPm, nlayers, gtape, dtape, P, ge, gd, fake, xn, yn)
        ctx.mark_non_differentiable(v, edge)
        return total, v, edge

    @staticmethod
    def backward(ctx, g_total, _gv, _ge):
        from .arch.Ours import networks as N
        method, GPm, nlayers, gtape, dtape, P, ge, gd, fake, xn, yn = ctx.state
        D = method.Discriminator
        B, dev = xn.shape[0], xn.device
        n = B * NPIX
        g_e = torch.empty((B, 1, 1, 1), dtype=torch.float32, device=dev)
        g_d = torch.empty((B, 64, 64, 1), dtype=torch.float32, device=dev)
        T = K.make_term
        K.loss_term_grads([T(0, ge, tconst=1.0, grad_out=g_e, coef=1.0 / B), T(0, gd, tconst=1.0, mx=xn, my=yn, grad_out=g_d, coef=1.0 / n)], dev)
        g_fake = DP.disc_backward(D._rt, P, dtape, g_e, g_d, None, None, True)        # input gradient only
        K.loss_term_grads([T(2, fake, yn, eps=method.pixel_loss.eps, grad_out=g_fake, coef=50.0 / n, accumulate=True)], dev)
        K.edge_loss(fake, yn, 50.0 / n, method.edge_loss.loss.eps, grad_out=g_fake, coef=50.0 / n, accumulate=True)
        g_up = g_total.detach().reshape(1).contiguous().float()
        if K.RECORDING is not None:
            K.RECORDING.keep.append(g_up)        # (autograd's tensor: a replay reads it again)
        g_fake = K.scale_by(g_fake, g_up)                                                # upstream scalar (== 1 from .backward())
        flat = N._flatten_gen(GPm)
        # One flat buffer, the parameter gradients are views of it (each on a 16-byte boundary): under data parallelism the
        # generator's gradients are then ONE in-place all-reduce on the RCCL stream (engine.train_iteration), like the
        # discriminator's task vectors -- no torch.cat, no 128 copies back.  (The views are handed to autograd and not kept
        # here: AccumulateGrad takes a gradient over as .grad without a copy only while nothing else refers to the tensor object.)
        sizes = [(p.numel() + 3) // 4 * 4 for p in flat]
        gbuf = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        gflat, ofs = [], 0
        for p, sz in zip(flat, sizes):
            gflat.append(gbuf[ofs:ofs + p.numel()].view_as(p))
            ofs += sz
        if ofs > sum(p.numel() for p in flat):
            K.zero_multi([gbuf])                            # the padding between views travels through the collective: keep it finite
        GP.generator_backward(g_fake, gtape, GPm, N._unflatten_gen(gflat, nlayers, as_grad=True))
        method._ggrad_flat = gbuf
        ctx.state = None
        return (None, None, None, None) + tuple(gflat)


def g_loss(method, x, y):
    G = method.Generator
    if not x.is_cuda:
        raise RuntimeError("MTD_GAN_Method.g_loss: HIP path needs CUDA tensors")
    # The D step's generator forward is reused only for the very tensor object it ran on (engine.train_iteration passes
    # the same x to d_loss and g_loss) with unchanged contents and weights; anything else -- a new batch of the same shape
    # at a recycled address, separate D and G batches, n_critic loops -- recomputes.
    gparams = G._flat_params()
    cache, method._gcache = getattr(method, "_gcache", None), None
    method._gcache_hit = cache if (cache is not None and cache[0] is x and cache[1] == _gkey(x, gparams)) else None
    total, v, edge = _GStepFn.apply(x, y, method, G._cfg[2], *gparams)
    method._gcache_hit = None
    details = {"G/gen_enc": v[0], "G/gen_dec": v[1], "G/pix_loss": v[2], "G/edge_loss": edge[0]}
    return total, details


# ================================================================================================ hipGraph replay
class GraphedTrainStep:
    """One full training iteration captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~3700
    kernel launches of a step cost one graph launch on the host.  Static shapes only (fixed batch); the inputs are
    copied into static buffers.  Host-side per-step state is refreshed before every replay: the PCGrad shuffle
    order (Python `random`, as the reference) and AdamW's step-dependent scalars; dropout masks come from the
    graph-safe Philox state.  Falls back to eager execution if capture is not possible (e.g. N > 1 with RCCL)."""

    def __init__(self, model, optimizer_G, optimizer_D, method_D, x, y, warmup=3):
        from . import engine
        self.model, self.oG, self.oD, self.wm = model, optimizer_G, optimizer_D, method_D
        self.x, self.y = x.clone(), y.clone()
        self.graph = None
        dev = x.device
        optimizer_G.graph_mode = optimizer_D.graph_mode = True     # step scalars through pinned slots (created in warm-up)
        K.prepare_capture(dev)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.names, self.vals = engine.train_iteration(model, self.x, self.y, optimizer_G, optimizer_D, method_D, None)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        K.CAPTURE_TAG += 1
        K.weights_changed(None)      # packed weight views made by eager steps must not be baked into the graph: the captured step packs its own
        model._gcache = None
        g = torch.cuda.CUDAGraph()
        rs = random.getstate()
        _pending_orders.append(shuffle_orders(3))        # placeholder order for the capture (capture executes nothing)
        random.setstate(rs)
        with torch.cuda.graph(g):
            self.names, self.vals = engine.train_iteration(model, self.x, self.y, optimizer_G, optimizer_D, method_D, None)
        # capture ran the host-side bookkeeping of a step that was not executed: take it back
        for opt in (optimizer_D, optimizer_G):
            for (_slot, _gi, members) in opt._captured:
                for p in members:
                    opt.state[p]["step"] -= 1
        self.graph = g
        self.slot = orders_slot(dev)
        self._done = None

    def step(self, x=None, y=None):
        # the graph's H2D copies read fixed pinned slots: do not rewrite them before the previous replay has run
        if self._done is not None:
            self._done.synchronize()
        if x is not None:
            self.x.copy_(x)
            self.y.copy_(y)
        orders = shuffle_orders(3)
        self.slot.set_inplace([j for o in orders for j in o])
        self.oD.advance_for_replay()
        self.oG.advance_for_replay()
        self.graph.replay()
        self._done = torch.cuda.Event()
        self._done.record()
        return self.names, self.vals


# ================================================================================================ launch-list replay
LIST_MODE = _options.product("MTD_LIST", "1") != "0"      # engine.train_MTD_GAN_Ours / bench: recorded launch list after two eager iterations
LIST_UNDER_DP = _options.product("MTD_LIST_DP", "1") != "0"   # ... also in a multi-rank group (0: data-parallel iterations stay eager)


class RecordingUnusable(RuntimeError):
    """The recorded iteration RAN TO COMPLETION (both optimizer steps applied) but the list it left cannot be replayed.
    Carries the iteration's (names, vals) so that the caller returns them and stays eager -- re-running the batch would
    apply it twice."""

    def __init__(self, why, names, vals):
        super().__init__(why)
        self.names, self.vals = names, vals


class RecordedTrainStep:
    """One full training iteration recorded as a kernels.LaunchList and re-issued per iteration: every C-ABI launch with the
    arguments it had, every stream-order operation (the weight-gradient side stream stays a side stream, unlike in a
    captured hipGraph, whose multi-stream sections ROCm 7.2 serialises), the uniform draws of the dropout masks and -- under
    data parallelism -- the RCCL collectives; none of the Python around them (tensor allocation, geometry and plan look-ups,
    the autograd engine).  The eager step costs the host ~24 ms per 30 ms of GPU work; the replay a fraction.

    The recording IS an iteration (it executes).  Host-side state of an iteration is refreshed per replay exactly as the
    eager step would produce it: the PCGrad projection order (Python's `random`, as reference module/weight_methods.py:452)
    and AdamW's step-dependent scalars go through pinned slots whose ring advances per replay; the dropout draws come from
    torch's generator at replay time.  Inputs are copied into the buffers the list reads.  Every tensor the recorded
    iteration allocated stays allocated (the list owns ~4 GB of activations and gradient buffers at 32 patches); parameter
    .grad tensors stay the recorded views, rewritten by every replay.

    Conditions (checked by `usable`; anything else runs eagerly): both optimizers are FusedAdamW, reduction 'sum', no
    injected dropout masks, fp32 CUDA inputs of the recorded shape, unchanged parameter storage and training mode."""

    def __init__(self, model, optimizer_G, optimizer_D, method_D, x, y, dp=None, warmup=2):
        from . import engine
        self.model, self.oG, self.oD, self.wm, self.dp = model, optimizer_G, optimizer_D, method_D, dp
        self.x, self.y = x.detach().float().contiguous().clone(), y.detach().float().contiguous().clone()
        dev = x.device
        optimizer_G.graph_mode = optimizer_D.graph_mode = True     # step scalars through pinned slots from now on (same kernel, same values)
        for _ in range(warmup):                                    # optimizer state, workspaces and derived weight views exist
            self.names, self.vals = engine.train_iteration(model, self.x, self.y, optimizer_G, optimizer_D, method_D, dp)
        self.list = K.LaunchList()
        try:
            self.names, self.vals = self.list.record(
                lambda: engine.train_iteration(model, self.x, self.y, optimizer_G, optimizer_D, method_D, dp), dev, repack=False)
        except RuntimeError as e:
            if self.list.result is None:       # the iteration itself failed: nothing to salvage
                raise
            raise RecordingUnusable(str(e), *self.list.result) from e
        try:
            self.slot = orders_slot(dev)
            if all(self.slot is not sl for sl in self.list.slots):
                raise RuntimeError("RecordedTrainStep: the recorded iteration did not read the PCGrad order slot")
        except RuntimeError as e:              # the iteration itself is done: do not let the caller run the batch again
            raise RecordingUnusable(str(e), self.names, self.vals) from e
        # the recorded AdamW launches hold raw pointers into the optimizers' moment tensors: keep those tensors alive with the
        # list, and let a replaced optimizer state (load_state_dict) retire it through the signature (advisor, round 4)
        self.moments = [t for opt in (optimizer_D, optimizer_G) for st in opt.state.values() for t in (st.get("exp_avg"), st.get("exp_avg_sq"))
                        if torch.is_tensor(t)]
        self.signature = self._signature(model, optimizer_G, optimizer_D, method_D, dp)
        self.grads = [(p, p.grad) for p in list(model.Discriminator.parameters()) + list(model.Generator.parameters())]
        self.iterations = warmup + 1

    @staticmethod
    def _signature(model, oG, oD, wm, dp):
        D, G = model.Discriminator, model.Generator
        # (version counters: an in-place torch update from outside -- load_state_dict, weight clipping, an EMA copy -- between two
        # iterations moves them; the list's own launches and FusedAdamW write through raw pointers and do not.  The list reads
        # derived weight views that only its own pack launches refresh, so such an update retires it: two eager iterations, then a
        # new recording.)
        return (tuple(p.data_ptr() for p in D.parameters()), tuple(p.data_ptr() for p in G.parameters()),
                tuple(b.data_ptr() for b in D.buffers()), D.training, G.training, D.c_drop.p,
                tuple(p.requires_grad for p in model.parameters()), id(oG), id(oD), id(wm), id(dp),
                tuple((g["betas"], g["eps"]) for g in oD.param_groups + oG.param_groups),
                tuple(t._version for t in model.parameters()), tuple(t._version for t in model.buffers()),
                tuple(t.data_ptr() for opt in (oD, oG) for st in opt.state.values() for t in (st.get("exp_avg"), st.get("exp_avg_sq"))
                      if torch.is_tensor(t)))

    @staticmethod
    def usable(model, optimizer_G, optimizer_D, method_D, x, y):
        """Can an iteration with these objects be recorded / replayed at all?"""
        from .optimizers import FusedAdamW
        from .module.weight_methods import PCGrad, WeightMethods
        D = getattr(model, "Discriminator", None)
        return (LIST_MODE and type(model).__name__ == "MTD_GAN_Method" and isinstance(optimizer_G, FusedAdamW) and isinstance(optimizer_D, FusedAdamW)
                and isinstance(method_D, WeightMethods) and isinstance(method_D.method, PCGrad) and method_D.method.reduction == "sum"
                and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape and tuple(x.shape[1:]) == (1, 64, 64)
                and not D._inject_masks and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing() and not POISON)

    def matches(self, model, optimizer_G, optimizer_D, method_D, x, y, dp):
        return (x.shape == self.x.shape and x.device == self.x.device and torch.cuda.current_stream() == self.list.main
                and self.signature == self._signature(model, optimizer_G, optimizer_D, method_D, dp)
                and not model.Discriminator._inject_masks)

    def step(self, x=None, y=None):
        """One iteration.  x, y: the batch (copied into the list's input buffers) or None to run on the resident batch."""
        if x is not None and x is not self.x:
            self.x.copy_(x)
            self.y.copy_(y)
        lst = self.list
        orders = shuffle_orders(3)                                  # this iteration's projection order, drawn as the reference draws it
        lst.set_slot(self.slot, [j for o in orders for j in o] + [0] * 7)
        self.oD.advance_for_replay(lst.set_slot)
        self.oG.advance_for_replay(lst.set_slot)
        lst.replay()
        self.orders = orders
        K.weights_changed(None)      # the parameters moved: nothing outside the list may trust a cached derived view
        if self.grads[0][0].grad is not self.grads[0][1]:           # someone cleared .grad (zero_grad): hand the recorded views back
            for p, g in self.grads:
                p.grad = g
        self.iterations += 1
        return self.names, self.vals


REPLICA_CHECKS = 2      # data-parallel replays whose result is compared across ranks (parallel.replicas_agree) before the list is trusted
# (each is a host synchronisation; two fall inside the five warm-up iterations of bench.py and of a training run's first seconds)


def _retire_list(model, optimizer_G, optimizer_D, why):
    """Stay eager for good with these objects, and say why once."""
    import warnings
    model._mtd_recorded = -(1 << 30)
    model._mtd_list_error = why
    optimizer_G.graph_mode = optimizer_D.graph_mode = False
    warnings.warn("mtd-gan_amd: the recorded launch list is not used (iterations stay eager): " + why, RuntimeWarning, stacklevel=3)


def recorded_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp=None):
    """engine.train_iteration through a RecordedTrainStep where one applies: the first two iterations of a configuration run
    eagerly, the third is recorded while it runs, later ones replay.  The step object lives on the model (it is tied to the
    model's parameter storage).  Returns (names, device tensor of the logged values) like engine.train_iteration.

    A recording that cannot be replayed never costs an iteration or a restart: the recorded iteration has run, its values are
    returned, and every later iteration is eager (`model._mtd_list_error` says why; bench.py reports it as `graph_error`).
    Under data parallelism the ranks decide that TOGETHER (one all-reduce of a flag), and the first REPLICA_CHECKS replays are
    followed by a comparison of the replicas' parameters across ranks: a list whose collectives were mis-ordered on real
    RCCL would show there, and is retired with rank 0's weights re-broadcast -- loudly, not silently."""
    from . import engine, parallel
    # (the list path needs the whole DataParallelSync protocol -- all_agree / replicas_agree / broadcast_* and collectives that go
    # through kernels.rec; a legacy hook object that only offers all_reduce_avg(flat) keeps its iterations eager)
    if (not RecordedTrainStep.usable(model, optimizer_G, optimizer_D, method_D, x, y)
            or (dp is not None and not (LIST_UNDER_DP and isinstance(dp, parallel.DataParallelSync)))):
        return engine.train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
    st = getattr(model, "_mtd_recorded", None)
    if isinstance(st, RecordedTrainStep):
        if st.matches(model, optimizer_G, optimizer_D, method_D, x, y, dp):
            out = st.step(x, y)
            if dp is not None and st.replica_checks_left > 0:
                st.replica_checks_left -= 1
                if not dp.replicas_agree(list(model.parameters()) + list(model.buffers())):
                    # the moments were updated from the same diverged gradients: rank 0's optimizer state goes with its weights
                    dp.broadcast_module(model)
                    dp.broadcast_optimizer_state((optimizer_D, optimizer_G))
                    _retire_list(model, optimizer_G, optimizer_D, "data-parallel replicas differed after a replayed iteration "
                                 "(rank 0's weights and AdamW moments were re-broadcast)")
                    model._mtd_eager_replica_checks = REPLICA_CHECKS      # ... and the eager iterations that follow are checked too
            return out
        if x.shape != st.x.shape:                                   # e.g. the last, smaller batch of an epoch: eager, keep the list
            return engine.train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
        model._mtd_recorded = st = None                             # different objects / storage: start over
    seen = (st or 0) if isinstance(st, int) else 0
    key = (tuple(x.shape), id(optimizer_G), id(optimizer_D))
    if getattr(model, "_mtd_recorded_key", None) != key:
        model._mtd_recorded_key, seen = key, 0
    if seen < 0:                                                    # retired
        out = engine.train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
        left = getattr(model, "_mtd_eager_replica_checks", 0)
        if dp is not None and left > 0:                             # a list was retired because the replicas had drifted: the repair is checked
            model._mtd_eager_replica_checks = left - 1
            if not dp.replicas_agree(list(model.parameters()) + list(model.buffers())):
                raise RuntimeError("mtd-gan_amd: data-parallel replicas differ after an EAGER iteration that followed a re-broadcast of rank 0's "
                                   "weights and optimizer state: the collectives of this process group do not deliver (not a launch-list problem)")
        return out
    if seen < 2:
        model._mtd_recorded = seen + 1
        return engine.train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
    why = None
    try:
        st = RecordedTrainStep(model, optimizer_G, optimizer_D, method_D, x, y, dp, warmup=0)
        out = (st.names, st.vals)
    except RecordingUnusable as e:                                  # the iteration ran; only the list is unusable
        st, out, why = None, (e.names, e.vals), str(e)
    except BaseException as e:
        # the iteration itself failed (or something other than a RuntimeError escaped the recording): nothing was completed that could be
        # returned, so the exception propagates -- but a caller that catches it must not re-record on every call with graph_mode left on,
        # and under data parallelism the peers are waiting in all_agree() below: tell them before leaving
        _retire_list(model, optimizer_G, optimizer_D, "the recorded iteration raised " + type(e).__name__)
        if dp is not None:
            try:
                dp.all_agree(False)
            except Exception:
                pass
        raise
    if dp is not None and not dp.all_agree(st is not None) and st is not None:
        st, why = None, "another rank could not record its iteration"
    if st is None:
        _retire_list(model, optimizer_G, optimizer_D, why)
        return out
    st.replica_checks_left = REPLICA_CHECKS if dp is not None else 0
    model._mtd_recorded = st
    model._mtd_list_error = None
    return out


# ================================================================================================ bench workload
class FullStepWorkload:
    """BASELINE config 3: full G + D + PCGrad training iteration (engine.train_MTD_GAN_Ours body) on
    32 patches per GPU; for N > 1 the per-task gradients are all-reduced before the projection so the
    result equals the single-process large-batch step."""

    name = "full_step"
    gflop_per_patch = 105.6

    def __init__(self, dev, rank, world, batch):
        from .arch.Ours.networks import MTD_GAN_Method
        from .data import synthetic_ldct
        from .module.weight_methods import WeightMethods
        from .optimizers import FusedAdamW
        from . import parallel
        torch.manual_seed(2024)
        random.seed(2024)
        self.model = MTD_GAN_Method().to(dev)
        self.model.train()
        x, y = synthetic_ldct(batch, seed=1234 + rank)
        self.x, self.y = x.to(dev), y.to(dev)
        self.world, self.batch = world, batch
        self.wm = WeightMethods("pcgrad", n_tasks=3, device=dev)
        self.oD = FusedAdamW(self.model.Discriminator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        self.oG = FusedAdamW(self.model.Generator.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
        import os
        force_dp = _options.product("MTD_FORCE_DP", "0") == "1"
        self.dp = parallel.DataParallelSync(dev) if (world > 1 or force_dp) else None
        if self.dp is not None:
            self.dp.broadcast_module(self.model)
            self.wm.method.dp = self.dp
        self.graphed = None
        self.graph_error = None
        self.recorded = None
        import os
        # hipGraph replay is opt-in (MTD_GRAPH=1): a captured step is slower than the launches it replaces, because ROCm 7.2
        # serialises the captured multi-stream sections; the default is the recorded launch list (MTD_LIST=0: eager)
        if world == 1 and _options.product("MTD_GRAPH", "0") == "1":
            try:
                self.graphed = GraphedTrainStep(self.model, self.oG, self.oD, self.wm, self.x, self.y)
            except Exception as e:                      # capture is an optimisation: report and run eagerly
                self.graph_error = repr(e)
                self.graphed = None
                torch.cuda.synchronize()

    def step(self):
        if self.graphed is not None:
            self.graphed.step()
        elif LIST_MODE:
            self.step_list()
        else:
            self.step_eager()

    def step_list(self):
        # the iteration as engine.train_MTD_GAN_Ours runs it by default: eager twice, recorded once, replayed from then on
        from . import engine
        if getattr(self, "_logged", None) is None:
            self._meters = {}
            self._logged = engine.LoggedScalars(self._meters, self.batch)
            engine.freeze_long_lived_objects(force=True)
        names, vals = recorded_iteration(self.model, self.x, self.y, self.oG, self.oD, self.wm, self.dp)
        self.recorded = getattr(self.model, "_mtd_recorded", None)
        self._logged.push(names, vals, self.oG.param_groups[0]["lr"])

    def step_eager(self):
        # one iteration as engine.train_MTD_GAN_Ours runs it: the schedule plus the logged scalars' lagged device -> host copy
        from . import engine
        if getattr(self, "_logged", None) is None:
            self._meters = {}
            self._logged = engine.LoggedScalars(self._meters, self.batch)
            engine.freeze_long_lived_objects(force=True)      # (process-wide gc.freeze(): opt-in for the API, MTD_GC_FREEZE=1)
        names, vals = engine.train_iteration(self.model, self.x, self.y, self.oG, self.oD, self.wm, self.dp)
        self._logged.push(names, vals, self.oG.param_groups[0]["lr"])

    def engine_api_ms(self, steps):
        """ms per iteration of `steps` iterations through the kept API itself, engine.train_MTD_GAN_Ours (same 9 positional
        arguments as reference engine.py:26), on a loader of `steps` resident batches -- what a caller of the reference's
        training loop sees, host-side logging included."""
        import time
        from . import engine
        loader = [dict(n_20=self.x, n_100=self.y)] * steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.train_MTD_GAN_Ours(self.model, loader, self.oG, self.oD, self.x.device, 0, 0, self.batch, self.wm)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    def config(self, world):
        return {"workload": "Full G+D+PCGrad train step (BASELINE configs[2]; configs[3] when N>1)", "per_gpu_batch": self.batch,
                "global_batch": self.batch * world, "patch": "1x64x64", "parallelism": f"dp{world}"}

    def extra(self):
        return {"algorithmic_gflop_per_patch": self.gflop_per_patch,
                "launch_mode": ("hipGraph replay" if getattr(self, "graphed", None) is not None else
                                "recorded launch list (side streams kept), replayed" if isinstance(getattr(self, "recorded", None), RecordedTrainStep)
                                else "eager launches"),
                "launches_per_step": (sum(1 for f, _a in self.recorded.list.ops if getattr(f, "__name__", "").startswith("mtd_"))
                                      if isinstance(getattr(self, "recorded", None), RecordedTrainStep) else None),
                "list_ops_per_step": len(self.recorded.list.ops) if isinstance(getattr(self, "recorded", None), RecordedTrainStep) else None,
                "graph_error": getattr(self, "graph_error", None) or getattr(self.model, "_mtd_list_error", None), "table_stats": dict(K.STATS)}

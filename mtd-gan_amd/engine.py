"""Mirror of the reference's step engine for MTD-GAN: engine.train_MTD_GAN_Ours (engine.py:26-76).

Same 9 positional arguments and the same returned dict {meter name: round(global average, 7)}.  The body
of one iteration is `train_iteration` (D step with PCGrad, then G step); the reference performs ~25 host
synchronisations per iteration (16 `.item()` in MetricLogger.update plus 9 `if dot < 0` in PCGrad), this
one gathers the 17 logged scalars into one device vector and copies it to the host once per iteration."""
from . import _options
import os

import torch


def train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp=None):
    """engine.py:38-55.  Returns (names, device tensor of the logged values)."""
    D, G = model.Discriminator, model.Generator
    # ---- Discriminator
    optimizer_D.zero_grad()
    D.zero_grad()
    d_losses, d_details = model.d_loss(x, y)
    if method_D is not None:
        method_D.backward(losses=d_losses, shared_parameters=list(D.shared_parameters()),
                          task_specific_parameters=list(D.task_specific_parameters()),
                          last_shared_parameters=list(D.last_shared_parameters()))
    else:
        d_losses.backward()                 # engine.py:56-63: the ablation wrappers return one scalar
        if dp is not None:
            dp.all_reduce_avg_list([p.grad for p in D.parameters() if p.grad is not None])
    optimizer_D.step()
    # ---- Generator
    optimizer_G.zero_grad()
    G.zero_grad()
    g_loss, g_details = model.g_loss(x, y)
    g_loss.backward()
    if dp is not None:
        _all_reduce_generator_grads(model, G, dp)
    optimizer_G.step()
    names = ["d_loss"] + list(d_details.keys()) + ["g_loss"] + list(g_details.keys())
    if d_losses.is_cuda:
        from . import kernels as K
        one = lambda t: (t.detach().reshape(-1).float().contiguous(), None)
        vals = K.scalar_sums([one(d_losses)] + [one(v) for v in d_details.values()] + [one(g_loss)] + [one(v) for v in g_details.values()],
                             d_losses.device)     # the 17 logged values in one launch (d_loss = the sum of the stacked task losses)
    else:       # (host tensors: the loop's own CPU tests drive it with stand-in models; the networks themselves refuse CPU tensors)
        vals = torch.stack([d_losses.detach().sum()] + [v.detach().reshape(()) for v in d_details.values()] + [g_loss.detach().reshape(())]
                           + [v.detach().reshape(()) for v in g_details.values()])
    return names, vals


def _all_reduce_generator_grads(model, G, dp):
    """The generator's gradients across ranks: one in-place collective on the flat buffer they are views of
    (train_step._GStepFn.backward), on the RCCL side stream; the generic bucket-and-scatter route for anything else."""
    flat = getattr(model, "_ggrad_flat", None)
    grads = [p.grad for p in G.parameters()]
    if flat is not None and all(g is not None and g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads):
        if hasattr(dp, "all_reduce_avg_inline"):
            dp.all_reduce_avg_inline(flat)          # on the current stream: the optimizer launch that follows needs it at once
        else:
            dp.all_reduce_avg(flat)
            dp.wait()
    else:
        dp.all_reduce_avg_list(grads)


class _Meter:
    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, value, n):
        self.total += value * n
        self.count += n

    @property
    def global_avg(self):
        return self.total / self.count


_gc_frozen = [False]


def freeze_long_lived_objects(force=False):
    """Collect, then move everything alive (modules, parameter tables, cached descriptors, the interpreter's own ~1e6 objects)
    to the garbage collector's permanent generation.  A full collection otherwise walks all of them every ~160 eager
    iterations -- 90-120 ms on the host (tools/hiccup_probe.py), three iterations' worth of GPU work.  Nothing is leaked that
    would have been freed: reference counting still frees frozen objects; only cycles among them are never looked for again.
    This changes PROCESS-WIDE interpreter state, so a library entry point does it only when asked: `MTD_GC_FREEZE=1` in the
    environment (train_MTD_GAN_Ours then calls it before its first iteration) or an explicit call with force=True (bench.py's
    workloads do, and say so in their line).  Once per process."""
    if _gc_frozen[0] or not (force or _options.product("MTD_GC_FREEZE", "0") == "1"):
        return
    import gc
    gc.collect()
    gc.freeze()
    _gc_frozen[0] = True


class LoggedScalars:
    """The iteration's ONE device -> host copy, taken off the host's critical path: the 17 logged values go to a pinned
    buffer with an asynchronous copy and are read one iteration later (or at once when a line is due to be printed), so the
    host keeps enqueuing the next iteration instead of waiting for this one to finish on the GPU.  The reference's
    MetricLogger.update calls `.item()` 16 times per iteration (utils.py:60-65)."""

    def __init__(self, meters, batch_size):
        self.meters, self.batch_size, self.pending = meters, batch_size, None
        self._ring, self._turn = [None, None], 0       # two pinned buffers, used in turn (one is pending while the other fills)

    def push(self, names, vals, lr):
        if vals.is_cuda:
            self._turn ^= 1
            host = self._ring[self._turn]
            if host is None or host.shape != vals.shape or host.dtype != vals.dtype:
                host = self._ring[self._turn] = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
            host.copy_(vals, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = vals, None
        self.drain()
        self.pending = (names, host, ev, lr)

    def drain(self):
        if self.pending is None:
            return
        names, host, ev, lr = self.pending
        self.pending = None
        if ev is not None:
            ev.synchronize()
        for k, v in zip(names, host.tolist()):
            self.meters.setdefault(k, _Meter()).update(v, self.batch_size)
        self.meters.setdefault("lr", _Meter()).update(lr, self.batch_size)


def train_MTD_GAN_Ours(model, data_loader, optimizer_G, optimizer_D, device, epoch, print_freq, batch_size, method_D, dp=None):
    """`dp`: optional data-parallel hook (parallel.DataParallelSync).  With PCGrad it defaults to the hook attached to the
    weight method (method_D.method.dp); the ablation wrappers' path (method_D=None) has no weight method to carry it, so a
    multi-process run passes it here -- and a process group of more than one rank without a hook is refused rather than
    left to train unsynchronised replicas."""
    model.Generator.train(True)
    model.Discriminator.train(True)
    if method_D is None and type(model).__name__ == "MTD_GAN_Method":
        raise NotImplementedError("the reference's method_D=None branch calls .backward() on MTD_GAN_Method's 3-vector and raises; "
                                  "use WeightMethods('pcgrad') (method_D=None is the ablation wrappers' path)")
    meters = {}
    if dp is None:
        dp = getattr(getattr(method_D, "method", None), "dp", None)
    if dp is None and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        raise RuntimeError("train_MTD_GAN_Ours: torch.distributed is initialised with more than one rank but no data-parallel hook was "
                           "given (pass dp=parallel.DataParallelSync(device), or attach it to the weight method): the ranks would "
                           "train unsynchronised replicas")
    n_it = len(data_loader)
    freeze_long_lived_objects()
    logged = LoggedScalars(meters, batch_size)
    from .train_step import recorded_iteration
    for it, batch_data in enumerate(data_loader):
        x = batch_data["n_20"].to(device).float()
        y = batch_data["n_100"].to(device).float()
        # (a recorded launch list replays the iteration from its third occurrence on where that applies -- train_step.
        # RecordedTrainStep: same launches, none of the Python around them; MTD_LIST=0 keeps every iteration eager)
        names, vals = recorded_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
        logged.push(names, vals, optimizer_G.param_groups[0]["lr"])
        if print_freq and (it % print_freq == 0 or it == n_it - 1):
            logged.drain()                            # a printed line shows this iteration, as the reference's does
            print(f"Train: [epoch:{epoch}] [{it}/{n_it}] " + "  ".join(f"{k}: {m.global_avg:.4f}" for k, m in meters.items()), flush=True)
    logged.drain()
    order = ["lr"] + [k for k in meters if k != "lr"]
    return {k: round(meters[k].global_avg, 7) for k in order}


# ================================================================================================ evaluation loops
def _l1(loss, pred, target):
    """The reference passes nn.L1Loss; its mean runs as one loss-term launch.  Any other callable is applied as given."""
    from . import kernels as K
    if isinstance(loss, torch.nn.L1Loss) and loss.reduction == "mean":
        p, t = pred.contiguous(), target.contiguous()
        return K.loss_terms([K.make_term(1, p, t, scale=1.0 / p.numel())], p.device)[0]
    return loss(pred, target)


def _save_png(path, img):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:                      # PNG dumps are cosmetic; the metrics do not depend on them
        return
    plt.imsave(path, img.squeeze().cpu().numpy(), cmap="gray")


@torch.no_grad()
def valid_MTD_GAN_Ours(model, loss, data_loader, device, epoch, save_dir, print_freq):
    """Mirror of engine.py:78-106: whole-slice generator inference + L1 loss; returns {'L1_loss': global average}.
    Slices may be 64, 128, 256 or 512 pixels square (the reference validates on 512 x 512)."""
    model.Generator.eval()
    model.Discriminator.eval()
    m = _Meter()
    n_it = len(data_loader)
    last = None
    for it, batch_data in enumerate(data_loader):
        x = batch_data["n_20"].to(device).float()
        y = batch_data["n_100"].to(device).float()
        pred = model.Generator(x)
        m.update(float(_l1(loss, pred, y)), 1)
        last = (x, y, pred)
        if print_freq and (it % print_freq == 0 or it == n_it - 1):
            print(f"Valid: [epoch:{epoch}] [{it}/{n_it}] L1_loss: {m.global_avg:.6f}", flush=True)
    if save_dir and last is not None:
        import os
        os.makedirs(save_dir, exist_ok=True)
        for tag, t in zip(("input_n_20", "gt_n_100", "pred_n_100"), last):
            _save_png(os.path.join(save_dir, f"epoch_{epoch}_{tag}.png"), t[0])
    return {"L1_loss": round(m.global_avg, 7)}


@torch.no_grad()
def test_MTD_GAN_Ours(model, loss, data_loader, device, save_dir):
    """Mirror of engine.py:108-183 for the pixel metrics: whole-slice inference, L1, RMSE / PSNR / SSIM of (input, gt,
    clipped prediction) per slice, pred_results.csv.  The perceptual metrics of the reference (PL, TML, FID: torchvision
    VGG16 / InceptionV3 weights) are outside this package; their columns are absent from the result."""
    from . import metrics as M
    model.Generator.eval()
    meters = {}
    rows = []
    for batch_data in data_loader:
        x = batch_data["n_20"].to(device).float()
        y = batch_data["n_100"].to(device).float()
        pred = model.Generator(x)
        meters.setdefault("L1_loss", _Meter()).update(float(_l1(loss, pred, y)), 1)
        from . import kernels as K
        pm = M.pixel_metrics(x, y, K.clip01(pred.contiguous()))
        for name, triple in pm.items():
            for who, v in zip(("input", "gt", "pred"), triple):
                meters.setdefault(f"{who}_{name}", _Meter()).update(v, 1)
        path = batch_data.get("path_n_20", [f"slice_{len(rows)}"])[0] if isinstance(batch_data, dict) else f"slice_{len(rows)}"
        rows.append((path, pm["rmse"][2], pm["psnr"][2], pm["ssim"][2]))
    if save_dir:
        import os
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "pred_results.csv"), "w") as f:
            f.write(",PATH,RMSE,PSNR,SSIM\n")
            for i, r in enumerate(rows):
                f.write(f"{i},{r[0]},{r[1]},{r[2]},{r[3]}\n")
    return {k: round(mm.global_avg, 7) for k, mm in meters.items()}

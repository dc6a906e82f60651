"""Mirror of the reference's step engine for MTD-GAN: engine.train_MTD_GAN_Ours (engine.py:26-76).

Same 9 positional arguments and the same returned dict {meter name: round(global average, 7)}.  The body
of one iteration is `train_iteration` (D step with PCGrad, then G step); the reference performs ~25 host
synchronisations per iteration (16 `.item()` in MetricLogger.update plus 9 `if dot < 0` in PCGrad), this
one gathers the 17 logged scalars into one device vector and copies it to the host once per iteration."""
import torch


def train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp=None):
    """engine.py:38-55.  Returns (names, device tensor of the logged values)."""
    D, G = model.Discriminator, model.Generator
    # ---- Discriminator
    optimizer_D.zero_grad()
    D.zero_grad()
    d_losses, d_details = model.d_loss(x, y)
    method_D.backward(losses=d_losses, shared_parameters=list(D.shared_parameters()),
                      task_specific_parameters=list(D.task_specific_parameters()),
                      last_shared_parameters=list(D.last_shared_parameters()))
    optimizer_D.step()
    # ---- Generator
    optimizer_G.zero_grad()
    G.zero_grad()
    g_loss, g_details = model.g_loss(x, y)
    g_loss.backward()
    if dp is not None:
        dp.all_reduce_avg_list([p.grad for p in G.parameters()])
    optimizer_G.step()
    names = ["d_loss"] + list(d_details.keys()) + ["g_loss"] + list(g_details.keys())
    vals = torch.stack([d_losses.sum()] + [v.reshape(()) for v in d_details.values()] + [g_loss.detach().reshape(())]
                       + [v.reshape(()) for v in g_details.values()])
    return names, vals


class _Meter:
    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, value, n):
        self.total += value * n
        self.count += n

    @property
    def global_avg(self):
        return self.total / self.count


def train_MTD_GAN_Ours(model, data_loader, optimizer_G, optimizer_D, device, epoch, print_freq, batch_size, method_D):
    model.Generator.train(True)
    model.Discriminator.train(True)
    if method_D is None:
        raise NotImplementedError("the reference's method_D=None branch calls .backward() on a 3-vector and raises; use WeightMethods('pcgrad')")
    meters = {}
    dp = getattr(getattr(method_D, "method", None), "dp", None)
    n_it = len(data_loader)
    for it, batch_data in enumerate(data_loader):
        x = batch_data["n_20"].to(device).float()
        y = batch_data["n_100"].to(device).float()
        names, vals = train_iteration(model, x, y, optimizer_G, optimizer_D, method_D, dp)
        host = vals.tolist()                          # the iteration's single device -> host copy
        for k, v in zip(names, host):
            meters.setdefault(k, _Meter()).update(v, batch_size)
        meters.setdefault("lr", _Meter()).update(optimizer_G.param_groups[0]["lr"], batch_size)
        if print_freq and (it % print_freq == 0 or it == n_it - 1):
            print(f"Train: [epoch:{epoch}] [{it}/{n_it}] " + "  ".join(f"{k}: {m.global_avg:.4f}" for k, m in meters.items()), flush=True)
    order = ["lr"] + [k for k in meters if k != "lr"]
    return {k: round(meters[k].global_avg, 7) for k in order}

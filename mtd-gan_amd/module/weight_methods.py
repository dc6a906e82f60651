"""Mirror of the reference's `module/weight_methods.py` for the method the MTD-GAN training loop uses:
WeightMethods('pcgrad') (reference :409-468 PCGrad, :727-761 WeightMethods facade / METHODS table).

backward(losses=..., shared_parameters=..., task_specific_parameters=..., last_shared_parameters=...)
writes `.grad` of the listed parameters and returns (None, {}) exactly like the reference.
  * losses produced by MTD_GAN_Method.d_loss carry the recorded discriminator passes: the fused per-task
    backward + Gram/combine kernels run (train_step.DStepTape.run_pcgrad);
  * any other autograd graph takes the generic route: one torch.autograd.grad per task, then the same
    HIP Gram / combine kernels on the flattened gradients.
The shuffle order is drawn from Python's `random` exactly as the reference does (one in-place shuffle of a
3-element list per task), so seeding `random` reproduces the reference's projection order.
"""
from typing import List, Union

import torch

from .. import kernels as K
from .. import train_step as TS


class WeightMethod:
    def __init__(self, n_tasks: int, device: torch.device):
        self.n_tasks = n_tasks
        self.device = device

    def parameters(self) -> List[torch.Tensor]:
        return []


class PCGrad(WeightMethod):
    def __init__(self, n_tasks: int, device: torch.device, reduction="sum"):
        super().__init__(n_tasks, device=device)
        assert reduction in ["mean", "sum"]
        self.reduction = reduction
        self.dp = None                 # optional data-parallel hook (parallel.DataParallelSync)

    def get_weighted_loss(self, losses, **kwargs):
        raise NotImplementedError

    def _set_pc_grads(self, losses, shared_parameters, task_specific_parameters=None):
        if isinstance(shared_parameters, torch.Tensor):
            shared_parameters = [shared_parameters]
        if isinstance(task_specific_parameters, torch.Tensor):
            task_specific_parameters = [task_specific_parameters]
        shared_parameters = list(shared_parameters)
        tape = getattr(losses, "_mtd_tape", None)
        if tape is not None and not tape.consumed:
            tape.run_pcgrad(shared_parameters, list(task_specific_parameters) if task_specific_parameters is not None else None,
                            self.reduction, self.dp)
            return
        # ---- generic autograd graph
        T = len(losses)
        if T > 4:
            raise NotImplementedError("HIP PCGrad kernels handle up to 4 tasks")
        sizes = [p.numel() for p in shared_parameters]
        flat = []
        for i in range(T):
            gs = torch.autograd.grad(losses[i], shared_parameters, retain_graph=True)
            flat.append(torch.cat([g.reshape(-1) for g in gs]).contiguous())
        orders = TS.shuffle_orders(T)
        orders_dev = torch.tensor([j for o in orders for j in o], dtype=torch.int32).to(flat[0].device, non_blocking=True)
        gram = K.pcgrad_gram(flat)
        merged = torch.empty_like(flat[0])
        K.pcgrad_combine(flat, gram, orders_dev, merged)
        if self.reduction == "mean":
            merged = merged / self.n_tasks
        ofs = 0
        for p, sz in zip(shared_parameters, sizes):
            p.grad = merged[ofs:ofs + sz].view_as(p)
            ofs += sz
        if task_specific_parameters is not None:
            task_specific_parameters = list(task_specific_parameters)
            ts = torch.autograd.grad(losses.sum(), task_specific_parameters)
            for p, g in zip(task_specific_parameters, ts):
                p.grad = g

    def backward(self, losses, parameters=None, shared_parameters=None, task_specific_parameters=None, **kwargs):
        self._set_pc_grads(losses, shared_parameters, task_specific_parameters)
        return None, {}          # NOTE: aligned with the reference (weight_methods.py:466-468)

    def __call__(self, losses, **kwargs):
        return self.backward(losses, **kwargs)


class WeightMethods:
    def __init__(self, method: str, n_tasks: int, device: torch.device, **kwargs):
        assert method in METHODS, f"unknown method {method}."
        self.method = METHODS[method](n_tasks=n_tasks, device=device, **kwargs)

    def get_weighted_loss(self, losses, **kwargs):
        return self.method.get_weighted_loss(losses, **kwargs)

    def backward(self, losses, **kwargs):
        return self.method.backward(losses, **kwargs)

    def __ceil__(self, losses, **kwargs):
        return self.backward(losses, **kwargs)

    def parameters(self):
        return self.method.parameters()


# the MTD-GAN training path uses PCGrad only (train.py:36,119-121); the reference's other weightings
# (NashMTL, MGDA, CAGrad, ...) are outside this build's scope
METHODS = dict(pcgrad=PCGrad)

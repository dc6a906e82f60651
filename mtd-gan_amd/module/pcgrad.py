"""Mirror of the reference's `module/pcgrad.py::PCGrad` (pcgrad.py:13-141): gradient surgery as an optimizer
wrapper -- `PCGrad(optimizer, reduction).pc_backward(objectives)`, `.step()`, `.zero_grad()`, `.optimizer`.

The reference's training loop does not use this class (it goes through module/weight_methods.py); north_star names
the file, so it is provided on the same HIP reduction kernels (csrc/pcgrad.hip): one Gram pass over the T flat task
gradients, the replay of the sequential projections on the T x T Gram matrix in one device thread, and one axpy per
run of parameters with the same reduction.  Differences from weight_methods.PCGrad that this class keeps, as the
reference has them:
  * the flat vector covers ALL parameters of the optimizer, in param_groups order (pcgrad.py:117-141); a parameter an
    objective does not reach contributes zeros and a has-grad mask of 0 (`_retrieve_grad`);
  * the dot products of the projection run over that whole vector (pcgrad.py:50-58), with the same cumulative
    in-place `random.shuffle` of the task list per i;
  * parameters every objective reaches ("shared") get the MEAN of the projected gradients -- `if self._reduction:` at
    pcgrad.py:60 is truthy for 'mean' and for 'sum', so the 'sum' branch is dead -- all others their SUM (:68-69); a
    falsy reduction ends the process in the reference (`exit`), here it raises.
There is no host synchronisation: the sign tests of the projection happen on the device.
"""
import random

import torch

from .. import kernels as K


class PCGrad:
    def __init__(self, optimizer, reduction="mean"):
        self._optim, self._reduction = optimizer, reduction

    @property
    def optimizer(self):
        return self._optim

    def zero_grad(self):
        """pcgrad.py:22-27"""
        return self._optim.zero_grad(set_to_none=True)

    def step(self):
        """pcgrad.py:29-34"""
        return self._optim.step()

    # ------------------------------------------------------------------------------------------------------------
    def _params(self):
        return [p for group in self._optim.param_groups for p in group["params"]]

    def pc_backward(self, objectives):
        """pcgrad.py:36-48.  objectives: a list of scalar losses on an autograd graph.  Writes .grad of every parameter
        of the optimizer (zeros where no objective reaches it, as the reference does)."""
        if getattr(objectives, "_mtd_tape", None) is not None:
            raise TypeError("module.pcgrad.PCGrad works on autograd objectives; the stacked losses of MTD_GAN_Method.d_loss carry "
                            "recorded passes instead -- use module.weight_methods.WeightMethods('pcgrad').backward for them")
        objectives = list(objectives)
        T = len(objectives)
        if not 1 <= T <= 4:
            raise NotImplementedError("HIP PCGrad kernels handle 1..4 objectives")
        if not self._reduction:
            raise ValueError("invalid reduction method")            # pcgrad.py:66: exit('invalid reduction method')
        params = self._params()
        if not params:
            return
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("module.pcgrad.PCGrad: HIP path needs CUDA parameters (there is no CPU fallback)")
        # flat layout: every parameter starts on a 16-byte boundary (float4 traffic in the kernels); the gaps stay zero
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros((T, max(total, 4)), dtype=torch.float32, device=dev)
        has = []                                                       # per objective, per parameter: reached or not
        for i, obj in enumerate(objectives):                           # pcgrad.py:96-103 (_pack_grad)
            self._optim.zero_grad(set_to_none=True)
            obj.backward(retain_graph=True)
            row = []
            for p, o in zip(params, offs):
                row.append(p.grad is not None)
                if p.grad is not None:
                    flat[i, o:o + p.numel()].copy_(p.grad.reshape(-1))
            has.append(row)
        shared = [all(h[k] for h in has) for k in range(len(params))]  # pcgrad.py:51
        # pcgrad.py:52-58 as a function of the Gram matrix of the ORIGINAL gradients
        orders = _shuffle_orders(T)
        vecs = [flat[i] for i in range(T)]
        gram = K.pcgrad_gram(vecs)
        od = torch.tensor([j for o in orders for j in o], dtype=torch.int32).to(dev, non_blocking=True)
        coeff = K.pcgrad_coeff(gram, od, T)
        merged = torch.empty(flat.shape[1], dtype=torch.float32, device=dev)
        # pcgrad.py:60-69: runs of consecutive parameters with the same reduction share one launch
        k = 0
        while k < len(params):
            e = k
            while e + 1 < len(params) and shared[e + 1] == shared[k]:
                e += 1
            lo, hi = offs[k], (offs[e + 1] if e + 1 < len(params) else total)
            K.pcgrad_axpy([v[lo:hi] for v in vecs], coeff, (1.0 / T) if shared[k] else 1.0, merged[lo:hi])
            k = e + 1
        for p, o in zip(params, offs):                                 # pcgrad.py:72-83 (_set_grad)
            p.grad = merged[o:o + p.numel()].view(p.shape)
        self.gram, self.coeff, self.orders, self.shared = gram, coeff, orders, shared


def _shuffle_orders(T):
    """Index orders of the reference's cumulative in-place `random.shuffle(grads)` per i (pcgrad.py:54): the same
    consumption of Python's `random` stream."""
    idx = list(range(T))
    out = []
    for _ in range(T):
        random.shuffle(idx)
        out.append(list(idx))
    return out

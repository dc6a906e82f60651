"""Fused multi-tensor AdamW with torch.optim.AdamW's update rule as the reference wires it
(train.py:122-126, optimizers.py:8-9: lr, betas (0.9, 0.999), eps 1e-8, weight_decay 5e-4, no amsgrad).
One HIP launch per step() over every parameter that has a gradient; parameters whose .grad is None are
skipped exactly like torch.optim (this is what keeps the reference's `c_fc` frozen)."""
import math

import torch

from . import kernels as K


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.graph_mode = False   # True: step-dependent scalars are read from pinned slots (hipGraph capture / replay)
        self._dyn = {}            # (group index, launch index) -> HostScalars with (decay, step_size, 1/sqrt(bc2))
        self._captured = []       # launches of the last step(): (HostScalars, group index, parameters)

    def load_state_dict(self, state_dict):
        """Accepts the optimizer section of a reference checkpoint (train.py:146-159 saves torch.optim.AdamW's state
        dict; its `step` entries are 0-d tensors since torch 1.12): steps become Python ints, moments stay tensors."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())
        for group in self.param_groups:                      # torch.optim.AdamW-only switches of a reference checkpoint
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdamW implements the reference's configuration (no amsgrad, no maximize)")

    def state_dict(self):
        """Same layout as torch.optim.AdamW's (train.py:276-288 saves it): `step` leaves as a 0-d float tensor, so that a
        checkpoint written with this optimizer resumes under the reference's torch.optim.AdamW as well."""
        sd = super().state_dict()
        state = {}
        for k, st in sd["state"].items():                     # the packed entries alias the live state: copy before editing
            st = dict(st)
            if isinstance(st.get("step"), int):
                st["step"] = torch.tensor(float(st["step"]))
            state[k] = st
        sd["state"] = state
        return sd

    @staticmethod
    def _scalars(group, step):
        b1, b2 = group["betas"]
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        return (1.0 - group["lr"] * group["weight_decay"], group["lr"] / bc1, 1.0 / math.sqrt(bc2))

    def advance_for_replay(self, set_slot=None):
        """Host-side bookkeeping of one replayed step: bump the step counters of the parameters the captured / recorded step
        updated and refresh the scalars its kernels read from pinned memory.  A hipGraph replay rewrites the one slot the
        graph is pinned to (its owner waits for the previous replay first); a launch-list replay passes
        set_slot = LaunchList.set_slot, which moves on in the slot's ring and re-points the recorded launch."""
        for (slot, gi, members) in self._captured:
            group = self.param_groups[gi]
            step = 0
            for p in members:
                st = self.state[p]
                st["step"] += 1
                step = st["step"]
            if set_slot is not None:
                set_slot(slot, self._scalars(group, step))
            else:
                slot.set_inplace(self._scalars(group, step))

    @torch.no_grad()
    def step(self, closure=None):
        self._captured = []
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdamW: HIP path needs CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(st["step"], []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            for step, items in by_step.items():
                b1, b2 = group["betas"]
                if not self.graph_mode:
                    # eager launch: the step-dependent scalars travel as kernel arguments
                    K.adamw_multi_pre([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items],
                                      b1, b2, group["eps"], self._scalars(group, step))
                    continue
                # hipGraph capture: kernel arguments are frozen, the scalars are read from a pinned slot instead
                slot = self._dyn.get((gi, len(self._captured)))
                if slot is None:
                    slot = K.HostScalars(items[0][0].device, 3, torch.float32)
                    self._dyn[(gi, len(self._captured))] = slot
                slot.set(self._scalars(group, step))
                self._captured.append((slot, gi, [it[0] for it in items]))
                K.adamw_multi_dyn([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items],
                                  b1, b2, group["eps"], slot.device_ptr())
                slot.consumed()
        # parameters were updated through raw pointers: their packed weight views (and only theirs) are stale
        K.weights_changed([p for group in self.param_groups for p in group["params"] if p.grad is not None])


def get_optimizer(name, model, lr):
    """reference optimizers.py:4-14 ('adamw' is what train.py uses for both networks)."""
    if name == "adamw":
        return FusedAdamW(model.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    raise KeyError(name)

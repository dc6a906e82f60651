"""Mirror of the reference's losses.py hot-path entries (losses.py:10-15 ls_gan / NDS_Loss, :99-111
CharbonnierLoss, :113-138 EdgeLoss) on the HIP loss kernels.  Inputs are NCHW (or (B,1)) CUDA tensors.
Each is a small autograd node so that foreign code can compose them; the fused training step
(train_step.py) batches the same kernels without going through autograd."""
import torch

from . import kernels as K


class _TermFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, kind, b, tconst, diffs, scale, eps):
        a_c = a.contiguous()
        b_c = b.contiguous() if b is not None else None
        mx = diffs.contiguous() if diffs is not None else None
        my = torch.zeros_like(mx) if mx is not None else None
        t = K.make_term(kind, a_c, b_c, tconst, mx, my, scale, eps)
        out = K.loss_terms([t], a.device)
        ctx.save = (a_c, b_c, mx, my, kind, tconst, scale, eps)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a, b, mx, my, kind, tconst, scale, eps = ctx.save
        ga = torch.empty_like(a)
        K.loss_term_grads([K.make_term(kind, a, b, tconst, mx, my, scale, eps, grad_out=ga, coef=scale)], a.device)
        ga = ga * g                                    # upstream scalar (plumbing)
        gb = -ga if (b is not None and ctx.needs_input_grad[2]) else None
        return ga, None, gb, None, None, None, None


def ls_gan(inputs, targets):
    """mean((inputs - targets)^2), targets a python scalar (losses.py:10-11)."""
    return _TermFn.apply(inputs, 0, None, float(targets), None, 1.0 / inputs.numel(), 0.0)


def NDS_Loss(inputs, targets, diffs):
    """mean(bool(|diffs|) * (inputs - targets)^2) over ALL elements (losses.py:13-15)."""
    return _TermFn.apply(inputs, 0, None, float(targets), diffs, 1.0 / inputs.numel(), 0.0)


def l1_loss(a, b):
    return _TermFn.apply(a, 1, b, 0.0, None, 1.0 / a.numel(), 0.0)


def mse_loss(a, b):
    return _TermFn.apply(a, 0, b, 0.0, None, 1.0 / a.numel(), 0.0)


class _ClipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        return K.clip01(xc)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        return K.clip01_bwd(g.contiguous(), xc)


def clip01(x):
    """x.clip(0, 1) with torch's gradient (1 inside [0, 1], boundaries included; 0 outside): networks.py:1750-1751,
    :1824-1825 (the restoration-consistency passes of the ablation wrappers)."""
    return _ClipFn.apply(x)


class CharbonnierLoss(torch.nn.Module):
    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return _TermFn.apply(x, 2, y, 0.0, None, 1.0 / x.numel(), self.eps)


class _EdgeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        xc, yc = x.contiguous(), y.contiguous()
        B = xc.shape[0]
        out = K.edge_loss(xc.reshape(B, 64, 64, 1), yc.reshape(B, 64, 64, 1), 1.0 / xc.numel(), eps)
        ctx.save = (xc, yc, eps)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        xc, yc, eps = ctx.save
        B = xc.shape[0]
        gx = torch.empty_like(xc)
        K.edge_loss(xc.reshape(B, 64, 64, 1), yc.reshape(B, 64, 64, 1), 1.0 / xc.numel(), eps, grad_out=gx, coef=1.0 / xc.numel())
        gx = gx * g
        return gx, (-gx if ctx.needs_input_grad[1] else None), None


class EdgeLoss(torch.nn.Module):
    """Charbonnier loss between Laplacian-pyramid residuals (5x5 Gaussian, replicate padding)."""

    def __init__(self):
        super().__init__()
        k = torch.Tensor([[.05, .25, .4, .25, .05]])
        self.kernel = torch.matmul(k.t(), k).unsqueeze(0).repeat(1, 1, 1, 1)      # kept for attribute parity
        self.loss = CharbonnierLoss()

    def forward(self, x, y):
        if tuple(x.shape[1:]) != (1, 64, 64):
            raise NotImplementedError("EdgeLoss HIP kernel: (B,1,64,64) patches")
        return _EdgeFn.apply(x, y, self.loss.eps)

"""MI355X-native mirror of the reference's `arch/Ours/networks.py` hot-path surface.

Same class names, constructor arguments, attribute names and state_dict keys as the reference
(arch/Ours/networks.py:15-36 FFT_ConvBlock, :38-164 ResFFT_Generator, :166-175 UpsampleBlock, :177-474
Multi_Task_Discriminator_Skip, :1940-2009 MTD_GAN_Method), so `from arch.Ours.networks import *`
call sites (models.py:15,52-53) can be pointed here.  Tensors are NCHW fp32 at this surface, exactly
like the reference; internally everything is NHWC and every arithmetic op is a hand-written gfx950
kernel from libmtdgan_hip.so.  There is no eager/CPU fallback: a CPU tensor or a missing library
raises.
"""
from itertools import chain
from typing import Iterator

import torch
import torch.nn as nn

from ... import generator_path as GP


def _require_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: this implementation runs on MI355X HIP kernels only; got a {t.device} tensor "
                           "(there is no CPU fallback -- use the reference for CPU runs)")


# =================================================================================================
# Res-FFT-Conv block
# =================================================================================================
class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_img, b_img, w_fft, b_fft):
        # x: NHWC contiguous (B,64,64,32)
        need = any(ctx.needs_input_grad)
        out, saved = GP.block_forward(x, w_img, b_img, w_fft, b_fft, need)
        if need:
            ctx.saved = saved
            ctx.w = (w_img, w_fft)
        return out

    @staticmethod
    def backward(ctx, g):
        w_img, w_fft = ctx.w
        grads = {"dw_img": torch.empty_like(w_img), "db_img": torch.empty(w_img.shape[0], device=g.device),
                 "dw_fft": torch.empty_like(w_fft), "db_fft": torch.empty(w_fft.shape[0], device=g.device)}
        gx = GP.block_backward(g.contiguous(), ctx.saved, w_img, w_fft, grads, False)
        K.side_stream(g.device).join()
        return gx, grads["dw_img"], grads["db_img"], grads["dw_fft"], grads["db_fft"]


class FFT_ConvBlock(nn.Module):
    """x + relu(conv3x3(x)) + irfft2(relu(conv1x1([Re;Im] rfft2(x))))  -- reference networks.py:15-36."""

    def __init__(self, out_channels):
        super().__init__()
        self.img_conv = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.fft_conv = nn.Conv2d(out_channels * 2, out_channels * 2, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        _require_cuda(x, "FFT_ConvBlock")
        if x.shape[1] != 32 or x.shape[2] != 64 or x.shape[3] != 64:
            raise NotImplementedError("FFT_ConvBlock HIP path: 32 channels, 64x64 patches (training hot path)")
        xn = x.permute(0, 2, 3, 1).contiguous()                 # layout plumbing only
        out = _BlockFn.apply(xn, self.img_conv.weight, self.img_conv.bias, self.fft_conv.weight, self.fft_conv.bias)
        return out.permute(0, 3, 1, 2)


# =================================================================================================
# Generator
# =================================================================================================
class _GeneratorFn(torch.autograd.Function):
    """Whole generator as one autograd node: forward and backward are explicit kernel schedules."""

    @staticmethod
    def forward(ctx, x, nlayers, *params):
        P = _unflatten_gen(params, nlayers)
        need = any(ctx.needs_input_grad)
        xn = x.reshape(x.shape[0], x.shape[2], x.shape[3], 1)    # C == 1: NCHW and NHWC coincide
        out, tape = GP.generator_forward(xn, P, need)
        if need:
            ctx.tape, ctx.P, ctx.nlayers = tape, P, nlayers
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        P, nlayers = ctx.P, ctx.nlayers
        flat = _flatten_gen(P)
        gflat = [torch.empty_like(p) for p in flat]
        G = _unflatten_gen(gflat, nlayers, as_grad=True)
        gn = g.contiguous().reshape(g.shape[0], g.shape[2], g.shape[3], 1)
        GP.generator_backward(gn, ctx.tape, P, G)
        ctx.tape = None
        return (None, None) + tuple(gflat)


def _flatten_gen(P):
    flat = []
    for w, b in zip(P.enc_w, P.enc_b):
        flat += [w, b]
    for w, b in zip(P.dec_w, P.dec_b):
        flat += [w, b]
    for blk in P.blk:
        flat += list(blk)
    return flat


def _unflatten_gen(flat, nlayers, as_grad=False):
    n = nlayers + 1
    enc_w, enc_b = list(flat[0:2 * n:2]), list(flat[1:2 * n:2])
    dec_w, dec_b = list(flat[2 * n:4 * n:2]), list(flat[2 * n + 1:4 * n:2])
    rest = flat[4 * n:]
    blk = []
    for i in range(0, len(rest), 4):
        if as_grad:
            blk.append({"dw_img": rest[i], "db_img": rest[i + 1], "dw_fft": rest[i + 2], "db_fft": rest[i + 3]})
        else:
            blk.append((rest[i], rest[i + 1], rest[i + 2], rest[i + 3]))
    return GP.GenParams(enc_w, enc_b, dec_w, dec_b, blk)


class ResFFT_Generator(nn.Module):
    """Reference networks.py:38-164.  RED-CNN-style 11 conv + 11 conv-transpose (stride 1) with additive
    skips and 21 Res-FFT-Conv blocks.  HIP path covers the MTD-GAN configuration (1, 32, 10, 3, 1)."""

    def __init__(self, in_channels=1, out_channels=96, num_layers=10, kernel_size=5, padding=0):
        super().__init__()
        encoder = [nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding)]
        decoder = [nn.ConvTranspose2d(out_channels, in_channels, kernel_size=kernel_size, stride=1, padding=padding)]
        for _ in range(num_layers):
            encoder.append(nn.Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding))
            decoder.append(nn.ConvTranspose2d(out_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding))
        self.encoder = nn.ModuleList(encoder)
        self.decoder = nn.ModuleList(decoder)
        self.enforce = nn.ModuleList([FFT_ConvBlock(out_channels) for _ in range(21)])
        self._cfg = (in_channels, out_channels, num_layers, kernel_size, padding)
        self.__init_weights()

    def __init_weights(self):
        # reference quirk: only Conv2d / Linear are re-initialised, ConvTranspose2d keeps
        # PyTorch's default init
        for m in self.modules():
            if type(m) in {nn.Conv2d, nn.Linear}:
                m.weight.data.normal_(0, 0.01)
                if hasattr(m.bias, "data"):
                    m.bias.data.fill_(0)

    def shared_parameters(self) -> Iterator[nn.parameter.Parameter]:
        return chain(*[self.encoder[i].parameters() for i in range(11)],
                     *[self.decoder[-i].parameters() for i in range(1, 12)])

    def task_specific_parameters(self):
        return None

    def last_shared_parameters(self) -> Iterator[nn.parameter.Parameter]:
        return self.decoder[-11].parameters()

    def _flat_params(self):
        flat = []
        for m in self.encoder:
            flat += [m.weight, m.bias]
        for m in self.decoder:
            flat += [m.weight, m.bias]
        for blk in self.enforce:
            flat += [blk.img_conv.weight, blk.img_conv.bias, blk.fft_conv.weight, blk.fft_conv.bias]
        return flat

    def forward(self, x: torch.Tensor):
        _require_cuda(x, "ResFFT_Generator")
        if self._cfg != (1, 32, 10, 3, 1):
            raise NotImplementedError("ResFFT_Generator HIP path is built for MTD_GAN_Method's (1,32,10,3,1) configuration")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != x.shape[3] or x.shape[2] not in (64, 128, 256, 512):
            raise NotImplementedError(f"ResFFT_Generator HIP path expects (B,1,S,S) with S in 64/128/256/512, got {tuple(x.shape)}")
        if x.shape[2] != 64 and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("ResFFT_Generator HIP path: maps larger than 64 x 64 are inference-only (use torch.no_grad())")
        if not torch.is_grad_enabled():
            xc = x.contiguous().float()
            P = _unflatten_gen(self._flat_params(), self._cfg[2])
            out, _ = GP.generator_forward(xc.reshape(xc.shape[0], xc.shape[2], xc.shape[3], 1), P, False)
            return out.reshape(xc.shape)
        return _GeneratorFn.apply(x.contiguous().float(), self._cfg[2], *self._flat_params())


# =================================================================================================
# Discriminator
# =================================================================================================
from ... import discriminator_path as DP  # noqa: E402
from ... import kernels as K  # noqa: E402
from ...losses import CharbonnierLoss, EdgeLoss, NDS_Loss, ls_gan  # noqa: E402


class UpsampleBlock(nn.Module):
    """1x1 conv to C*scale^2 channels + PixelShuffle -- reference networks.py:166-175."""

    def __init__(self, scale, input_channels, output_channels):
        super().__init__()
        self.upsample = nn.Sequential(
            nn.Conv2d(input_channels, output_channels * (scale ** 2), kernel_size=1, stride=1, padding=0),
            nn.PixelShuffle(upscale_factor=scale))

    def forward(self, input):
        """Stand-alone use (inside Multi_Task_Discriminator_Skip the same two steps run in place in the decoder's concatenated
        buffer, discriminator_path.disc_forward): conv1x1 on the implicit GEMM, PixelShuffle(2) as its own kernel."""
        _require_cuda(input, "UpsampleBlock")
        conv, ps = self.upsample[0], self.upsample[1]
        if ps.upscale_factor != 2:
            raise NotImplementedError("UpsampleBlock HIP path: PixelShuffle(2) only (the only scale the reference uses, networks.py:267-301)")
        if input.dim() != 4 or input.shape[1] != conv.in_channels:
            raise RuntimeError(f"UpsampleBlock: expected (B, {conv.in_channels}, H, W), got {tuple(input.shape)}")
        if conv.in_channels % 32 or conv.out_channels % 32:
            raise NotImplementedError("UpsampleBlock HIP path: channel counts in multiples of 32 (the implicit-GEMM and weight-gradient "
                                      "kernels' tiles; the reference's six instances are 512 / 256 / 128 / 64 -> the same)")
        return _UpsampleFn.apply(input.contiguous().float(), conv.weight, conv.bias)


class _UpsampleFn(torch.autograd.Function):
    """conv1x1 (C -> 4 C') + PixelShuffle(2) and its autograd transpose on the HIP kernels (reference networks.py:166-175)."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, Ci, H, W = x.shape
        C4 = w.shape[0]
        xn = x.permute(0, 2, 3, 1).contiguous()                              # NHWC
        up = K.empty_nhwc(B, H, W, C4, xn)
        K.conv(xn, w.detach(), K.geom_fwd(B, H, W, 1, 1, 0), C4, Ci, Ci, 1, up, bias=b.detach() if b is not None else None)
        out = K.empty_nhwc(B, 2 * H, 2 * W, C4 // 4, xn)
        K.pixel_shuffle2_fwd(up, out)
        ctx.save_for_backward(xn, w)
        ctx.has_bias = b is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        xn, w = ctx.saved_tensors
        B, H, W, Ci = xn.shape
        C4 = w.shape[0]
        gn = g.permute(0, 2, 3, 1).contiguous().float()
        gr = K.empty_nhwc(B, H, W, C4, xn)
        K.pixel_shuffle2_bwd(gn, gr)
        gq = K.geom_fwd(B, H, W, 1, 1, 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gxn = K.empty_nhwc(B, H, W, Ci, xn)
            K.conv(gr, w.detach(), gq, Ci, C4, 1, Ci, gxn)                  # W^T: W(n = ci, c = co) = w[co*Ci + ci]
            gx = gxn.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.zeros_like(w)
            gb = torch.zeros(C4, dtype=torch.float32, device=w.device) if ctx.has_bias else None
            K.wgrad(gr, xn, gq, C4, Ci, gw, Ci, 1, db=gb)
        return gx, gw, gb


class _DiscFn(torch.autograd.Function):
    """One whole discriminator pass as an autograd node (generic use; the training step drives the same
    schedules directly, see train_step.py)."""

    @staticmethod
    def forward(ctx, x, module, train, mask, need_rec, *params):
        ctx.set_materialize_grads(False)
        names = module._param_names
        P = dict(zip(names, params))
        P.update(module._buffer_dict())
        xn = x.reshape(x.shape[0], 64, 64, 1)
        need = any(ctx.needs_input_grad)
        (enc, dec, rec), tape = DP.disc_forward(P, xn, train, mask, need_rec, need)
        if need:
            ctx.tape, ctx.P, ctx.module, ctx.names = tape, P, module, names
        B = x.shape[0]
        outs = (enc.reshape(B, 1), dec.reshape(B, 1, 64, 64), rec.reshape(B, 1, 64, 64) if rec is not None else None)
        if rec is None:
            ctx.mark_non_differentiable()
        return outs

    @staticmethod
    def backward(ctx, g_enc, g_dec, g_rec):
        B = ctx.tape.B
        names, P = ctx.names, ctx.P
        needs = ctx.needs_input_grad[5:]
        sink_t = {n: torch.zeros_like(P[n]) for n, need in zip(names, needs) if need}
        f = lambda g, shape: g.contiguous().reshape(shape) if g is not None else None
        gin = DP.disc_backward(ctx.module._rt, P, ctx.tape, f(g_enc, (B, 1, 1, 1)), f(g_dec, (B, 64, 64, 1)), f(g_rec, (B, 64, 64, 1)),
                               DP.GradSink(sink_t) if sink_t else None, ctx.needs_input_grad[0])
        K.side_stream(P[names[0]].device).join()
        gx = gin.reshape(B, 1, 64, 64) if gin is not None else None
        return (gx, None, None, None, None) + tuple(sink_t.get(n) for n in names)


class Multi_Task_Discriminator_Skip(nn.Module):
    """Reference networks.py:177-474: spectral-norm conv trunk 64x64 -> 1x1 (skips x1..x6), CLS head,
    bilinear SEG decoder, PixelShuffle REC decoder.  Attribute / parameter / buffer names follow the
    reference (`<layer>.bias`, `.weight_orig`, `.weight_u`, `.weight_v`, `r_up{k}.upsample.0.*`)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        if (in_channels, out_channels) != (1, 64):
            raise NotImplementedError("HIP path is built for MTD_GAN_Method's (1, 64) configuration")
        sn = nn.utils.spectral_norm
        c = out_channels
        chans = [c, c * 2, c * 4, c * 8, c * 8, c * 8]
        cin = in_channels
        for l, co in enumerate(chans, start=1):
            setattr(self, f"conv{l}1", sn(nn.Conv2d(cin, co, 3, 1, 1)))
            setattr(self, f"relu{l}1", nn.LeakyReLU(0.2))
            setattr(self, f"conv{l}2", sn(nn.Conv2d(co, co, 3, 1, 1)))
            setattr(self, f"relu{l}2", nn.LeakyReLU(0.2))
            setattr(self, f"down{l}", sn(nn.Conv2d(co, co, 4, 2, 1)))
            cin = co
        self.bconv1 = sn(nn.Conv2d(c * 8, c * 8, 1, 1, 0))
        self.brelu1 = nn.LeakyReLU(0.2)
        self.bconv2 = sn(nn.Conv2d(c * 8, c * 8, 1, 1, 0))
        self.brelu2 = nn.LeakyReLU(0.2)
        self.c_flatten = nn.Flatten()
        self.c_fc = sn(nn.Linear(512, 512, True))
        self.c_relu = nn.LeakyReLU(0.2)
        self.c_drop = nn.Dropout(p=0.3)
        for pre in ("s", "r"):
            for l, (ci, co) in enumerate(DP.DEC, start=1):
                if pre == "s":
                    setattr(self, f"s_up{l}", nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
                else:
                    setattr(self, f"r_up{l}", UpsampleBlock(2, DP.RUP[l - 1][0], DP.RUP[l - 1][1]))
                setattr(self, f"{pre}_dconv{l}1", sn(nn.Conv2d(ci, co, 3, 1, 1)))
                setattr(self, f"{pre}_drelu{l}1", nn.LeakyReLU(0.2))
                setattr(self, f"{pre}_dconv{l}2", sn(nn.Conv2d(co, co, 3, 1, 1)))
                setattr(self, f"{pre}_drelu{l}2", nn.LeakyReLU(0.2))
        self.enc_out = nn.Linear(512, 1)
        self.dec_out = nn.Conv2d(in_channels, 1, 1)
        self.rec_out = nn.Conv2d(in_channels, 1, 1)
        self.__init_weights()
        self._rt = DP.DiscRuntime()
        self._param_names = [n for n, _ in self.named_parameters()]
        self._inject_masks = []          # tests: dropout multipliers (B,512) consumed one per forward

    def __init_weights(self):
        for m in self.modules():
            if type(m) in {nn.Conv2d, nn.Linear}:
                m.weight.data.normal_(0, 0.01)
                if hasattr(m.bias, "data"):
                    m.bias.data.fill_(0)

    # ---- parameter partition (reference networks.py:318-380; `c_fc` is in neither list) ----------
    def shared_parameters(self) -> Iterator[nn.parameter.Parameter]:
        names = [f"{k}{l}{j}" if k == "conv" else f"down{l}" for l in range(1, 7) for k, j in (("conv", 1), ("conv", 2), ("down", ""))]
        return chain(*[getattr(self, n).parameters() for n in names], self.bconv1.parameters(), self.bconv2.parameters())

    def task_specific_parameters(self) -> Iterator[nn.parameter.Parameter]:
        mods = [getattr(self, f"s_dconv{l}{j}") for l in range(1, 7) for j in (1, 2)]
        for l in range(1, 7):
            mods += [getattr(self, f"r_up{l}"), getattr(self, f"r_dconv{l}1"), getattr(self, f"r_dconv{l}2")]
        mods += [self.enc_out, self.dec_out, self.rec_out]
        return chain(*[m.parameters() for m in mods])

    def last_shared_parameters(self) -> Iterator[nn.parameter.Parameter]:
        return self.bconv2.parameters()

    # ---- plumbing ------------------------------------------------------------------------------------
    def _buffer_dict(self):
        return {n: b for n, b in self.named_buffers()}

    def _param_dict(self):
        P = {n: p for n, p in self.named_parameters()}
        P.update(self._buffer_dict())
        return P

    def _next_mask(self, B, device):
        return self._next_masks(B, device, 1)

    def _next_masks(self, B, device, count):
        """The dropout multipliers (networks.py:313 c_drop, p = 0.3) of the next `count` forward passes of B images, stacked
        along the batch: (count * B, 512), or None in eval mode / for p = 0.  The uniform draws are torch's generator (so
        `torch.manual_seed` governs them as it governs nn.Dropout), one draw for all `count` passes; thresholding and the
        1 / (1 - p) scale are a library launch (mtd_dropout_mask) -- both recordable by kernels.LaunchList."""
        if not self.training:
            return None
        if self._inject_masks:                              # tests: recorded masks of the reference, one per pass
            return torch.cat([self._inject_masks.pop(0).to(device) for _ in range(count)], 0)
        p = self.c_drop.p
        if p == 0.0:
            return None
        r = torch.empty((count * B, 512), dtype=torch.float32, device=device)
        K.rec(r.uniform_)                                   # RNG draw only
        return K.dropout_mask(r, p, torch.empty_like(r))

    def forward(self, input, need_rec=True):
        _require_cuda(input, "Multi_Task_Discriminator_Skip")
        if input.dim() != 4 or tuple(input.shape[1:]) != (1, 64, 64):
            raise NotImplementedError(f"discriminator expects (B,1,64,64) (Linear(512,512) after the 1x1 bottleneck), got {tuple(input.shape)}")
        x = input.contiguous().float()
        mask = self._next_mask(x.shape[0], x.device)
        params = [p for _, p in self.named_parameters()]
        return _DiscFn.apply(x, self, self.training, mask, need_rec, *params)


# =================================================================================================
# Ablation family (reference networks.py:478-1937): RED-CNN generator, the five partial discriminators
# and the ten wrappers.  Same kernels and schedules as above -- the discriminators are the trunk of
# Multi_Task_Discriminator_Skip with a subset of its heads (discriminator_path.disc_forward(heads=...)).
# The wrappers' d_loss / g_loss return ONE scalar (engine.train_MTD_GAN_Ours runs them with
# method_D=None: plain .backward(), engine.py:56-73), so they are composed from the autograd nodes of
# this module and of losses.py.  Not mirrored: the reference's print() of the score maxima in every
# d_loss / g_loss (a host synchronisation per call).
# =================================================================================================
class _RedcnnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nlayers, *params):
        n = nlayers + 1
        enc_w, enc_b = list(params[0:2 * n:2]), list(params[1:2 * n:2])
        dec_w, dec_b = list(params[2 * n:4 * n:2]), list(params[2 * n + 1:4 * n:2])
        need = any(ctx.needs_input_grad)
        xn = x.reshape(x.shape[0], x.shape[2], x.shape[3], 1)
        out, tape = GP.redcnn_forward(xn, enc_w, enc_b, dec_w, dec_b, need)
        if need:
            ctx.tape, ctx.w = tape, (enc_w, dec_w, params)
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        enc_w, dec_w, params = ctx.w
        n = len(enc_w)
        grads = [torch.empty_like(p) for p in params]
        gn = g.contiguous().reshape(g.shape[0], g.shape[2], g.shape[3], 1)
        GP.redcnn_backward(gn, ctx.tape, enc_w, dec_w, grads[0:2 * n:2], grads[1:2 * n:2], grads[2 * n:4 * n:2], grads[2 * n + 1:4 * n:2])
        ctx.tape = None
        return (None, None) + tuple(grads)


class REDCNN_Generator(nn.Module):
    """Reference networks.py:478-505: 11 conv + 11 conv-transpose (stride 1), additive skips from every encoder INPUT,
    no Res-FFT blocks; every Conv* layer is re-initialised N(0, 0.01) (unlike ResFFT_Generator, whose ConvTranspose2d
    layers keep PyTorch's default init).  HIP path: the ablation wrappers' configuration (1, 32, 10, 3, 1), 64 x 64 patches."""

    def __init__(self, in_channels=1, out_channels=96, num_layers=10, kernel_size=5, padding=0):
        super().__init__()
        encoder = [nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding)]
        decoder = [nn.ConvTranspose2d(out_channels, in_channels, kernel_size=kernel_size, stride=1, padding=padding)]
        for _ in range(num_layers):
            encoder.append(nn.Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding))
            decoder.append(nn.ConvTranspose2d(out_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding))
        self.encoder = nn.ModuleList(encoder)
        self.decoder = nn.ModuleList(decoder)
        self._cfg = (in_channels, out_channels, num_layers, kernel_size, padding)
        self.__init_weights()

    def __init_weights(self):
        for m in self.modules():
            if m.__class__.__name__.find("Conv") != -1:
                m.weight.data.normal_(0, 0.01)
                if hasattr(m.bias, "data"):
                    m.bias.data.fill_(0)

    def _flat_params(self):
        flat = []
        for m in self.encoder:
            flat += [m.weight, m.bias]
        for m in self.decoder:
            flat += [m.weight, m.bias]
        return flat

    def forward(self, x: torch.Tensor):
        _require_cuda(x, "REDCNN_Generator")
        if self._cfg != (1, 32, 10, 3, 1):
            raise NotImplementedError("REDCNN_Generator HIP path is built for the ablation wrappers' (1,32,10,3,1) configuration")
        if x.dim() != 4 or tuple(x.shape[1:]) != (1, 64, 64):
            raise NotImplementedError(f"REDCNN_Generator HIP path expects (B,1,64,64) patches, got {tuple(x.shape)}")
        return _RedcnnFn.apply(x.contiguous().float(), self._cfg[2], *self._flat_params())


class _PartialDiscFn(torch.autograd.Function):
    """One pass of a partial discriminator as an autograd node (the generic twin of _DiscFn: parameter names are mapped
    to Multi_Task_Discriminator_Skip's, absent heads give None)."""

    @staticmethod
    def forward(ctx, x, module, train, mask, *params):
        ctx.set_materialize_grads(False)
        names = module._canon_names
        P = dict(zip(names, params))
        P.update({module._canon(n): b for n, b in module.named_buffers()})
        xn = x.reshape(x.shape[0], 64, 64, 1)
        need = any(ctx.needs_input_grad)
        (enc, dec, rec), tape = DP.disc_forward(P, xn, train, mask, True, need, heads=module.HEADS)
        if need:
            ctx.tape, ctx.P, ctx.module, ctx.names = tape, P, module, names
        B = x.shape[0]
        return (enc.reshape(B, 1) if enc is not None else None, dec.reshape(B, 1, 64, 64) if dec is not None else None,
                rec.reshape(B, 1, 64, 64) if rec is not None else None)

    @staticmethod
    def backward(ctx, g_enc, g_dec, g_rec):
        B = ctx.tape.B
        names, P = ctx.names, ctx.P
        needs = ctx.needs_input_grad[4:]
        sink_t = {n: torch.zeros_like(P[n]) for n, need in zip(names, needs) if need and n not in ctx.module._unused}
        f = lambda g, shape: g.contiguous().reshape(shape) if g is not None else None
        gin = DP.disc_backward(ctx.module._rt, P, ctx.tape, f(g_enc, (B, 1, 1, 1)), f(g_dec, (B, 64, 64, 1)), f(g_rec, (B, 64, 64, 1)),
                               DP.GradSink(sink_t) if sink_t else None, ctx.needs_input_grad[0])
        K.side_stream(P[names[0]].device).join()
        gx = gin.reshape(B, 1, 64, 64) if gin is not None else None
        return (gx, None, None, None) + tuple(sink_t.get(n) for n in names)


class _PartialDiscriminator(nn.Module):
    """Trunk of Multi_Task_Discriminator_Skip + the heads in HEADS, attribute names as in the reference class.
    SEG_PREFIX: '' for SEG_Discriminator (up1, dconv11, ...: networks.py:611-697), 's_' otherwise."""

    HEADS = ()
    SEG_PREFIX = "s_"
    EXTRA_ENC_OUT = False          # SEG_Discriminator carries an enc_out Linear that its forward never uses (networks.py:695)

    def __init__(self, in_channels, out_channels):
        super().__init__()
        if (in_channels, out_channels) != (1, 64):
            raise NotImplementedError("HIP path is built for the ablation wrappers' (1, 64) configuration")
        sn = nn.utils.spectral_norm
        c = out_channels
        cin = in_channels
        for l, co in enumerate([c, c * 2, c * 4, c * 8, c * 8, c * 8], start=1):
            setattr(self, f"conv{l}1", sn(nn.Conv2d(cin, co, 3, 1, 1)))
            setattr(self, f"relu{l}1", nn.LeakyReLU(0.2))
            setattr(self, f"conv{l}2", sn(nn.Conv2d(co, co, 3, 1, 1)))
            setattr(self, f"relu{l}2", nn.LeakyReLU(0.2))
            setattr(self, f"down{l}", sn(nn.Conv2d(co, co, 4, 2, 1)))
            cin = co
        self.bconv1 = sn(nn.Conv2d(c * 8, c * 8, 1, 1, 0))
        self.brelu1 = nn.LeakyReLU(0.2)
        self.bconv2 = sn(nn.Conv2d(c * 8, c * 8, 1, 1, 0))
        self.brelu2 = nn.LeakyReLU(0.2)
        if "cls" in self.HEADS:
            self.c_flatten = nn.Flatten()
            self.c_fc = sn(nn.Linear(512, 512, True))
            self.c_relu = nn.LeakyReLU(0.2)
            self.c_drop = nn.Dropout(p=0.3)
        for head, pre in (("seg", self.SEG_PREFIX), ("rec", "r_")):
            if head not in self.HEADS:
                continue
            for l, (ci, co) in enumerate(DP.DEC, start=1):
                if head == "seg":
                    setattr(self, f"{pre}up{l}", nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
                else:
                    setattr(self, f"{pre}up{l}", UpsampleBlock(2, DP.RUP[l - 1][0], DP.RUP[l - 1][1]))
                setattr(self, f"{pre}dconv{l}1", sn(nn.Conv2d(ci, co, 3, 1, 1)))
                setattr(self, f"{pre}drelu{l}1", nn.LeakyReLU(0.2))
                setattr(self, f"{pre}dconv{l}2", sn(nn.Conv2d(co, co, 3, 1, 1)))
                setattr(self, f"{pre}drelu{l}2", nn.LeakyReLU(0.2))
        if "cls" in self.HEADS or self.EXTRA_ENC_OUT:
            self.enc_out = nn.Linear(512, 1)
        if "seg" in self.HEADS:
            self.dec_out = nn.Conv2d(in_channels, 1, 1)
        if "rec" in self.HEADS:
            self.rec_out = nn.Conv2d(in_channels, 1, 1)
        self.__init_weights()
        self._rt = DP.DiscRuntime()
        self._canon_names = [self._canon(n) for n, _ in self.named_parameters()]
        self._unused = {"enc_out.weight", "enc_out.bias"} if (self.EXTRA_ENC_OUT and "cls" not in self.HEADS) else set()
        self._inject_masks = []

    def __init_weights(self):
        for m in self.modules():
            if type(m) in {nn.Conv2d, nn.Linear}:
                m.weight.data.normal_(0, 0.01)
                if hasattr(m.bias, "data"):
                    m.bias.data.fill_(0)

    def _canon(self, name):
        """This class's parameter / buffer name -> Multi_Task_Discriminator_Skip's (the schedules' vocabulary)."""
        if self.SEG_PREFIX == "" and name.startswith("dconv"):
            return "s_" + name
        return name

    def _next_mask(self, B, device):
        if not self.training or "cls" not in self.HEADS:
            return None
        if self._inject_masks:
            return self._inject_masks.pop(0).to(device)
        p = self.c_drop.p
        if p == 0.0:
            return None
        return (torch.rand(B, 512, device=device) >= p).to(torch.float32) / (1.0 - p)

    def _run(self, input):
        _require_cuda(input, type(self).__name__)
        if input.dim() != 4 or tuple(input.shape[1:]) != (1, 64, 64):
            raise NotImplementedError(f"discriminator expects (B,1,64,64), got {tuple(input.shape)}")
        x = input.contiguous().float()
        mask = self._next_mask(x.shape[0], x.device)
        return _PartialDiscFn.apply(x, self, self.training, mask, *[p for _, p in self.named_parameters()])


class CLS_Discriminator(_PartialDiscriminator):
    """networks.py:507-609: trunk + image-level head; returns x_enc."""
    HEADS = ("cls",)

    def forward(self, input):
        return self._run(input)[0]


class SEG_Discriminator(_PartialDiscriminator):
    """networks.py:611-764: trunk + bilinear pixel-level decoder (attributes up{l} / dconv{l}{j}, no prefix); returns x_dec."""
    HEADS = ("seg",)
    SEG_PREFIX = ""
    EXTRA_ENC_OUT = True

    def forward(self, input):
        return self._run(input)[1]


class CLS_SEG_Discriminator(_PartialDiscriminator):
    """networks.py:766-932; returns (x_enc, x_dec)."""
    HEADS = ("cls", "seg")

    def forward(self, input):
        e, d, _ = self._run(input)
        return e, d


class CLS_REC_Discriminator(_PartialDiscriminator):
    """networks.py:934-1101; returns (x_enc, x_rec)."""
    HEADS = ("cls", "rec")

    def forward(self, input):
        e, _, r = self._run(input)
        return e, r


class SEG_REC_Discriminator(_PartialDiscriminator):
    """networks.py:1103-1322; returns (x_dec, x_rec)."""
    HEADS = ("seg", "rec")

    def forward(self, input):
        _, d, r = self._run(input)
        return d, r


def _l1(a, b):
    from ...losses import l1_loss
    return l1_loss(a, b)


def _mse(a, b):
    from ...losses import mse_loss
    return mse_loss(a, b)


class _AblationBase(nn.Module):
    """Shared body of the ten ablation wrappers (networks.py:1324-1937).  Class attributes say which generator and
    discriminator the wrapper owns and which terms its losses have; the formulas below are the reference's, term by term,
    including its quirks (Ablation_CLS_REC's generator loss scores the RESTORATION output with ls_gan and logs it as
    'G/gen_dec', networks.py:1520-1538)."""

    GEN = "redcnn"               # or "resfft"
    DISC = None                  # discriminator class
    OUTS = ()                    # what the discriminator returns, in order: "enc", "dec", "rec"
    NDS = False                  # pixel-level adversarial terms through NDS_Loss(x - y) instead of ls_gan
    RC = False                   # restoration-consistency terms

    def __init__(self):
        super().__init__()
        self.Generator = (REDCNN_Generator if self.GEN == "redcnn" else ResFFT_Generator)(in_channels=1, out_channels=32, num_layers=10,
                                                                                       kernel_size=3, padding=1)
        self.Discriminator = self.DISC(in_channels=1, out_channels=64)
        if self.NDS:
            self.gan_metric_cls = ls_gan
            self.gan_metric_seg = NDS_Loss
        else:
            self.gan_metric = ls_gan
        self.pixel_loss = CharbonnierLoss()
        self.edge_loss = EdgeLoss()

    def _d(self, t):
        o = self.Discriminator(t)
        o = o if isinstance(o, tuple) else (o,)
        return dict(zip(self.OUTS, o))

    def _cls(self, t, target):
        return ls_gan(t, target)

    def _seg(self, t, target, x, y):
        return NDS_Loss(t, target, x - y) if self.NDS else ls_gan(t, target)

    def d_loss(self, x, y):
        from ...losses import clip01
        fake = self.Generator(x).detach()
        real, fk = self._d(y), self._d(fake)
        details = {}
        if "enc" in real:
            details["D/real_enc"], details["D/fake_enc"] = self._cls(real["enc"], 1.0), self._cls(fk["enc"], 0.0)
        if "dec" in real:
            details["D/real_dec"], details["D/fake_dec"] = self._seg(real["dec"], 1.0, x, y), self._seg(fk["dec"], 0.0, x, y)
        if self.ENC_DEC_ORDER_INTERLEAVED:      # (summation order of the reference's expression, for bit-level agreement of the total)
            order = ["D/real_enc", "D/real_dec", "D/fake_enc", "D/fake_dec"]
        else:
            order = ["D/real_enc", "D/fake_enc", "D/real_dec", "D/fake_dec"]
        terms = [details[k] for k in order if k in details]
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        # the reference recomputes every logged adversarial term (a second, identical kernel launch): same values
        if "rec" in real:
            details["D/rec_loss_real"] = _l1(real["rec"], y)
            details["D/rec_loss_fake"] = _l1(fk["rec"], fake)
            total = total + (details["D/rec_loss_real"] + details["D/rec_loss_fake"])
        if self.RC:
            rr, rf = self._d(clip01(real["rec"])), self._d(clip01(fk["rec"]))
            c = {"D/consist_loss_real_enc": _mse(real["enc"], rr["enc"]), "D/consist_loss_real_dec": _mse(real["dec"], rr["dec"]),
                 "D/consist_loss_fake_enc": _mse(fk["enc"], rf["enc"]), "D/consist_loss_fake_dec": _mse(fk["dec"], rf["dec"])}
            details.update(c)
            total = total + (((c["D/consist_loss_real_enc"] + c["D/consist_loss_real_dec"]) + c["D/consist_loss_fake_enc"])
                             + c["D/consist_loss_fake_dec"])
        return total, details

    ENC_DEC_ORDER_INTERLEAVED = False

    def g_loss(self, x, y):
        fake = self.Generator(x)
        gen = self._d(fake)
        details = {}
        adv = None
        for out_key, name in self.G_TERMS:
            # the reference's wrappers score whatever their discriminator returns FIRST / SECOND and log it as gen_enc / gen_dec
            # (Ablation_SEG_REC: the pixel-level map and the restoration; Ablation_CLS_REC: the score and the restoration)
            t = self._seg(gen[out_key], 1.0, x, y) if (out_key == "dec" and name == "G/gen_dec") else ls_gan(gen[out_key], 1.0)
            details[name] = t
            adv = t if adv is None else adv + t
        pix = 50.0 * self.pixel_loss(fake, y)
        edge = 50.0 * self.edge_loss(fake, y)
        details["G/pix_loss"], details["G/edge_loss"] = pix, edge
        return adv + pix + edge, details

    G_TERMS = ()


_G_ENC, _G_ENC_DEC = (("enc", "G/gen_enc"),), (("enc", "G/gen_enc"), ("dec", "G/gen_dec"))


class Ablation_CLS(_AblationBase):
    DISC, OUTS, G_TERMS = CLS_Discriminator, ("enc",), _G_ENC


class Ablation_SEG(_AblationBase):
    """networks.py:1374-1424: the pixel-level map is called *_enc throughout this wrapper (same keys as Ablation_CLS)."""
    DISC, OUTS, G_TERMS = SEG_Discriminator, ("enc",), _G_ENC


class Ablation_CLS_SEG(_AblationBase):
    DISC, OUTS, G_TERMS = CLS_SEG_Discriminator, ("enc", "dec"), _G_ENC_DEC
    ENC_DEC_ORDER_INTERLEAVED = True


class Ablation_CLS_REC(_AblationBase):
    DISC, OUTS, G_TERMS = CLS_REC_Discriminator, ("enc", "rec"), (("enc", "G/gen_enc"), ("rec", "G/gen_dec"))


class Ablation_SEG_REC(_AblationBase):
    DISC, OUTS, G_TERMS = SEG_REC_Discriminator, ("dec", "rec"), (("dec", "G/gen_enc"), ("rec", "G/gen_dec"))


class Ablation_CLS_SEG_REC(_AblationBase):
    DISC, OUTS, G_TERMS = Multi_Task_Discriminator_Skip, ("enc", "dec", "rec"), _G_ENC_DEC
    ENC_DEC_ORDER_INTERLEAVED = True


class Ablation_CLS_SEG_REC_NDS(_AblationBase):
    DISC, OUTS, G_TERMS, NDS = Multi_Task_Discriminator_Skip, ("enc", "dec", "rec"), _G_ENC_DEC, True


class Ablation_CLS_SEG_REC_RC(_AblationBase):
    DISC, OUTS, G_TERMS, RC = Multi_Task_Discriminator_Skip, ("enc", "dec", "rec"), _G_ENC_DEC, True
    ENC_DEC_ORDER_INTERLEAVED = True


class Ablation_CLS_SEG_REC_NDS_RC(_AblationBase):
    DISC, OUTS, G_TERMS, NDS, RC = Multi_Task_Discriminator_Skip, ("enc", "dec", "rec"), _G_ENC_DEC, True, True


class Ablation_CLS_SEG_REC_NDS_RC_ResFFT(_AblationBase):
    GEN = "resfft"
    DISC, OUTS, G_TERMS, NDS, RC = Multi_Task_Discriminator_Skip, ("enc", "dec", "rec"), _G_ENC_DEC, True, True


# =================================================================================================
# MTD-GAN (reference networks.py:1940-2009)
# =================================================================================================
class MTD_GAN_Method(nn.Module):
    def __init__(self):
        super().__init__()
        self.Generator = ResFFT_Generator(in_channels=1, out_channels=32, num_layers=10, kernel_size=3, padding=1)
        self.Discriminator = Multi_Task_Discriminator_Skip(in_channels=1, out_channels=64)
        self.gan_metric_cls = ls_gan
        self.gan_metric_seg = NDS_Loss
        self.pixel_loss = CharbonnierLoss()
        self.edge_loss = EdgeLoss()

    def d_loss(self, x, y):
        """Returns (stack[disc, rec, consist], details).  The stacked tensor carries the recorded
        discriminator passes (`_mtd_tape`), which WeightMethods('pcgrad').backward consumes."""
        from ... import train_step as TS
        return TS.d_loss(self, x, y)

    def g_loss(self, x, y):
        from ... import train_step as TS
        return TS.g_loss(self, x, y)

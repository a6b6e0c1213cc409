"""ContentLossDiscriminator on the HIP path.  Mirrors reference loss.py:231-289: constructor ``(config, criterion="mse")``,
``forward(x, gt) -> 0-dim``, taps / weights from ``config.MODEL.G_LOSS.DISC_FEATURES_LOSS_LAYERS`` on a freshly constructed
(the reference never loads trained weights into it) Discriminator in eval mode, ImageNet input normalisation.

Kernel graph (like vgg_loss.py): SR and GT go through the first discriminator layers as ONE batch of 2B (eval-mode BatchNorm is
a per-channel affine, applied together with LeakyReLU while the next conv stages its input); a tap is the conv output plus its
affine, the criterion kernel applies affine + LeakyReLU itself (csrc/misc.hip: feat_loss_*); backward runs on the SR half
only (activation/affine backward, stride-1 / stride-2 data-gradient convs), no weight gradients (frozen)."""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _abi, ops
from .model import Discriminator

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
LRELU = 0.2
# (conv idx in features, bn idx or None, stride); the LeakyReLU output of a layer is features.<(bn or conv) + 1>   (model.py:30-59)
_PLAN = [(0, None, 1), (2, 3, 2), (5, 6, 1), (8, 9, 2), (11, 12, 1), (14, 15, 2), (17, 18, 1), (20, 21, 2)]


def _feat_loss_fwd(a, b, scale, shift, C, mode, ws):
    n = a.numel()
    key = (a.device, n)
    if ws.get("key") != key:
        ws["key"] = key
        ws["partials"] = torch.empty(_abi.lib().sst_pixel_loss_blocks(n), device=a.device, dtype=torch.float32)
        ws["counter"] = torch.zeros(1, device=a.device, dtype=torch.int32)
    loss = torch.empty((), device=a.device, dtype=torch.float32)
    _abi.check(_abi.lib().sst_feat_loss_fwd(_abi.ptr(a), _abi.ptr(b), _abi.ptr(scale), _abi.ptr(shift), LRELU, C, _abi.ptr(loss),
                                            _abi.ptr(ws["partials"]), _abi.ptr(ws["counter"]), n, mode, _abi.stream_ptr()),
               "sst_feat_loss_fwd")
    return loss


class _DiscFeatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt, module, grad_mode):
        need = grad_mode and ctx.needs_input_grad[0]
        B = x.shape[0]
        feats = module.D.features
        xin = torch.cat([x, gt], dim=0).contiguous()
        h = ops.transpose_affine(xin, False, module.norm_scale, module.norm_shift)        # (x-mean)/std, NHWC
        wp = module.packed(0)
        saved, taps = [], {}
        scale = shift = None
        act = 0
        for ci, bi, stride in module.plan:
            conv = feats[ci]
            y = ops.conv_fwd(h, wp[ci], conv.weight.shape[0], 3, stride, bias=conv.bias, in_scale=scale, in_shift=shift,
                             in_slope_const=LRELU, in_act=act)[0]
            if bi is not None:
                bn = feats[bi]
                scale, shift = ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var)
            else:
                scale = shift = None
            saved.append((ci, stride, h, y, scale, shift))
            tap = (bi if bi is not None else ci) + 1
            if tap in module.tap_weights:
                taps[tap] = (y, scale, shift)
            h, act = y, ops.ACT_SLOPE
        terms = [_feat_loss_fwd(y[:B], y[B:], sc, sh, y.shape[-1], module.mode, module._ws.setdefault(t, {}))
                 for t, (y, sc, sh) in sorted(taps.items())]
        arr = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
        wts = (ctypes.c_float * len(terms))(*[float(module.tap_weights[t]) for t in sorted(taps)])
        out = torch.empty((), device=x.device, dtype=torch.float32)
        _abi.check(_abi.lib().sst_weighted_sum(arr, wts, len(terms), _abi.ptr(out), None, _abi.stream_ptr()), "sst_weighted_sum")
        if need:
            ctx.module, ctx.saved, ctx.taps, ctx.B = module, saved, taps, B
        return out

    @staticmethod
    def backward(ctx, gout):
        module, saved, taps, B = ctx.module, ctx.saved, ctx.taps, ctx.B
        feats = module.D.features
        wd = module.packed(1)
        gout = gout.contiguous()
        g = None                       # gradient w.r.t. the activated output of the current layer (SR half)
        lib = _abi.lib()
        for ci, stride, hin, y, scale, shift in reversed(saved):
            ysr = y[:B].contiguous()
            dy = None
            if g is not None:          # through LeakyReLU and the eval-mode BatchNorm affine: dy = scale * act'(z) * g
                dy = ops.bwd_apply(g, ysr, scale=scale, shift=shift, slope_const=LRELU, act=1, cA=scale,
                                   cB=module.zeros(scale) if scale is not None else None,
                                   cC=module.zeros(scale) if scale is not None else None)
            bi = next(b for c, b, _ in module.plan if c == ci)
            tap = (bi if bi is not None else ci) + 1
            if tap in taps:
                if dy is None:
                    dy = torch.empty_like(ysr)
                    acc = 0
                else:
                    acc = 1
                _abi.check(lib.sst_feat_loss_bwd(_abi.ptr(ysr), _abi.ptr(y[B:].contiguous()), _abi.ptr(scale), _abi.ptr(shift), LRELU,
                                                 ysr.shape[-1], _abi.ptr(dy), _abi.ptr(gout), float(module.tap_weights[tap]), acc,
                                                 ysr.numel(), module.mode, _abi.stream_ptr()), "sst_feat_loss_bwd")
            w = feats[ci].weight
            if stride == 1:
                g = ops.conv_fwd(dy, wd[ci], w.shape[1], 3, 1)[0]
            else:
                Hin, Win = hin.shape[1], hin.shape[2]
                g = ops.conv_s2_dgrad(dy, ops.pack_conv_s2_dgrad(w), Hin, Win, w.shape[1])
        dx = ops.transpose_affine(g, True, module.inv_std)
        ctx.saved = ctx.taps = None
        return dx, None, None, None


class ContentLossDiscriminator(nn.Module):
    def __init__(self, config, criterion: str = "mse") -> None:
        super().__init__()
        if criterion == "l1":
            self.mode = 1
        elif criterion in ("l2", "mse"):
            self.mode = 0
        else:
            raise NotImplementedError("%s criterion has not been implmented." % criterion)
        self.extraction_layers = dict(config.MODEL.G_LOSS.DISC_FEATURES_LOSS_LAYERS)
        self.device = config.DEVICE
        self.tap_weights = {int(k.split(".")[1]): float(v) for k, v in self.extraction_layers.items()}
        valid = {(b if b is not None else c) + 1 for c, b, _ in _PLAN}
        for t in self.tap_weights:
            if t not in valid:
                raise NotImplementedError(f"features.{t} is not a LeakyReLU output of the discriminator; only those taps are built")
        last = max(self.tap_weights)
        self.plan = [(c, b, s) for c, b, s in _PLAN if (b if b is not None else c) + 1 <= last]
        self.D = Discriminator(config)                       # loss.py:263: a fresh discriminator, never loaded from a checkpoint
        for p in self.D.parameters():
            p.requires_grad = False
        self.D.eval()
        self.register_buffer("norm_scale", torch.tensor([1.0 / s for s in STD]))
        self.register_buffer("norm_shift", torch.tensor([-m / s for m, s in zip(MEAN, STD)]))
        self.register_buffer("inv_std", torch.tensor([1.0 / s for s in STD]))
        self._packed, self._ws, self._zeros = {}, {}, {}
        self.to(self.device)

    def zeros(self, like):
        z = self._zeros.get(like.numel())
        if z is None or z.device != like.device:
            z = self._zeros[like.numel()] = torch.zeros_like(like)
        return z

    def packed(self, mode):
        convs = [(c, self.D.features[c].weight) for c, _, _ in self.plan]
        key = (mode, tuple((w.data_ptr(), w._version) for _, w in convs))
        if self._packed.get("key" + str(mode)) != key:
            self._packed["key" + str(mode)] = key
            self._packed[mode] = {c: ops.pack_conv(w, mode) for c, w in convs}
        return self._packed[mode]

    def forward(self, x, gt):
        if not x.is_cuda:
            raise _abi.HipPathError("ContentLossDiscriminator: the HIP path needs ROCm device tensors (no CPU fallback)")
        return _DiscFeatFn.apply(x, gt, self, torch.is_grad_enabled())

    def __repr__(self):
        return "ContentLossDiscriminator()"

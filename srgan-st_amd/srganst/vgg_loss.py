"""ContentLossVGG on the HIP path.  Mirrors reference loss.py:11-70 (constructor ``(config, criterion="mse")``,
``forward(x, gt) -> 0-dim``, taps / weights from ``config.MODEL.G_LOSS.VGG19_LAYERS``).

The reference downloads torchvision's IMAGENET1K_V1 weights (loss.py:46) - a network fetch that is impossible here.
``weights`` may be a path to a torchvision-format state dict (``features.N.weight|bias``; loaded with
``torch.load(weights_only=True)``); without it the stack is initialised like torchvision's VGG (seeded) so that
throughput and kernel parity can still be measured.  Frozen weights => no weight gradients, packed once.

Kernel graph: SR and GT go through the stack as ONE batch of 2B (the weights stream once); convs are the generic
fp32-MFMA kernel with ReLU applied on load, ReLU+MaxPool is one kernel, the feature criterion is the pixel-loss
kernel in relu mode; backward runs on the SR half only (dgrad convs, max-pool routing, ReLU masks)."""
from __future__ import annotations

import torch
from torch import nn

from . import _abi, ops

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _plan(upto):
    out, idx, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            out.append((idx, "pool", cin, cin))
            idx += 1
        else:
            out.append((idx, "conv", cin, v))
            out.append((idx + 1, "relu", v, v))
            idx += 2
            cin = v
    return [l for l in out if l[0] <= upto]


class _VggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt, module, grad_mode):
        need = grad_mode and ctx.needs_input_grad[0]
        B = x.shape[0]
        taps, weights = module.taps, module.tap_weights
        xin = torch.cat([x, gt], dim=0).contiguous()                                   # [2B,3,H,W]
        h = ops.transpose_affine(xin, False, module.norm_scale, module.norm_shift)     # (x-mean)/std, NHWC
        wp = module.packed(0)
        act, saved, feats = 0, [], {}
        plan = module.plan
        i = 0
        while i < len(plan):
            idx, kind, cin, cout = plan[i]
            assert kind == "conv"
            y = ops.conv_fwd(h, wp[idx], cout, 3, 1, bias=module.features[idx].bias, in_slope_const=0.0, in_act=act)[0]
            saved.append((idx, h, act, y))
            relu_idx = idx + 1
            if relu_idx in taps:
                feats[relu_idx] = y                                                    # pre-activation; relu applied by the criterion
            if i + 2 < len(plan) and plan[i + 2][1] == "pool":
                h, act = ops.maxpool_relu_fwd(y), 0
                saved.append((plan[i + 2][0], None, None, y))                          # pool marker: y is its (pre-act) input
                i += 3
            else:
                h, act = y, ops.ACT_SLOPE
                i += 2
        loss_terms, ws = [], module._ws
        total = None
        for t in taps:
            y = feats[t]
            n = y[:B].numel()
            l = ops.pixel_loss_fwd(y[:B], y[B:], module.mode | 2, ws.setdefault(t, {}))
            loss_terms.append(l)
        # weighted sum of the (<= 8) scalars in one tiny launch
        import ctypes
        arr = (ctypes.c_void_p * len(taps))(*[l.data_ptr() for l in loss_terms])
        wts = (ctypes.c_float * len(taps))(*[float(weights[t]) for t in taps])
        out = torch.empty((), device=x.device, dtype=torch.float32)
        _abi.check(_abi.lib().sst_weighted_sum(arr, wts, len(taps), _abi.ptr(out), None, _abi.stream_ptr()), "sst_weighted_sum")
        if need:
            ctx.module, ctx.saved, ctx.feats, ctx.B = module, saved, feats, B
        return out

    @staticmethod
    def backward(ctx, gout):
        module, saved, feats, B = ctx.module, ctx.saved, ctx.feats, ctx.B
        taps, weights = module.taps, module.tap_weights
        wd = module.packed(1)
        gout = gout.contiguous()
        g = None            # gradient wrt the activated output of the current layer (SR half)
        dx = None
        for rec in reversed(saved):
            idx, hin, act, y = rec
            ysr = y[:B]
            if hin is None:                                     # max-pool marker: route g to the arg-max of relu(y)
                dy_pool = ops.maxpool_relu_bwd(g, ysr)
                g = ("dy", dy_pool)                             # already masked by relu'
                continue
            # gradient wrt y (pre-activation of conv idx): relu mask on g (+ the feature-criterion gradient at a tap)
            relu_idx = idx + 1
            if isinstance(g, tuple):
                dy = g[1]
            elif g is not None:
                dy = ops.bwd_apply(g, ysr, slope_const=0.0, act=1)
            else:
                dy = None
            if relu_idx in taps:
                n = ysr.numel()
                dy = ops.pixel_loss_bwd(ysr, y[B:], module.mode | 2, scale_dev=gout, scale_host=float(weights[relu_idx]),
                                        out=dy, accumulate=dy is not None)
            cin = module.features[idx].weight.shape[1]
            g = ops.conv_fwd(dy, wd[idx], cin, 3, 1)[0]         # wrt the conv input (activated previous layer / pooled / normalised image)
        dx = ops.transpose_affine(g, True, module.inv_std)      # d/dx of (x-mean)/std, back to NCHW
        ctx.saved = ctx.feats = None
        return dx, None, None, None


class ContentLossVGG(nn.Module):
    def __init__(self, config, criterion: str = "mse", weights: str | None = None, seed: int = 0, allow_random: bool = False) -> None:
        """weights: path of a torchvision-format VGG19 state dict (keys ``features.{i}.weight|bias``; what the reference's
        ``models.vgg19(weights=IMAGENET1K_V1)`` holds, loss.py:46).  The ImageNet weights are a network fetch, so there is no
        default: without `weights` the constructor refuses unless the caller opts in to a seeded-random network with
        allow_random=True (throughput benchmarks and parity tests only - a perceptual term against random features trains
        nothing meaningful)."""
        super().__init__()
        if not weights and not allow_random:
            raise ValueError("ContentLossVGG: pass weights=<torchvision VGG19 state dict> (reference: IMAGENET1K_V1, loss.py:46) or "
                             "allow_random=True for a seeded-random VGG19 (benchmarks / tests only)")
        if criterion == "l1":
            self.mode = 1
        elif criterion in ("l2", "mse"):
            self.mode = 0
        else:
            raise NotImplementedError("%s criterion has not been implmented." % criterion)
        self.extraction_layers = dict(config.MODEL.G_LOSS.VGG19_LAYERS)
        self.device = config.DEVICE
        self.taps = sorted(int(k.split(".")[1]) for k in self.extraction_layers)
        self.tap_weights = {int(k.split(".")[1]): float(v) for k, v in self.extraction_layers.items()}
        self.plan = _plan(max(self.taps))
        for t in self.taps:
            if not any(i == t and k == "relu" for i, k, _, _ in self.plan):
                raise NotImplementedError(f"features.{t} is not a ReLU output of VGG19; only ReLU taps are built on the HIP path")
        layers = []
        for idx, kind, cin, cout in _plan(36):
            layers.append(nn.Conv2d(cin, cout, 3, 1, 1) if kind == "conv" else (nn.ReLU(True) if kind == "relu" else nn.MaxPool2d(2)))
        self.features = nn.Sequential(*layers)
        g = torch.Generator().manual_seed(seed)
        for m in self.features:
            if isinstance(m, nn.Conv2d):                        # torchvision's VGG init
                with torch.no_grad():
                    m.weight.normal_(0, (2.0 / (m.weight.shape[0] * 9)) ** 0.5, generator=g)
                    m.bias.zero_()
        if weights:
            sd = torch.load(weights, map_location="cpu", weights_only=True)
            own = self.state_dict()
            missing = [k for k in own if k.startswith("features.") and k not in sd]
            bad = [k for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
            if missing or bad:      # a wrong / partial / renamed state dict must not silently leave random layers behind
                raise ValueError(f"ContentLossVGG: {weights} is not a VGG19 `features` state dict: missing {missing[:4]}"
                                 f"{'...' if len(missing) > 4 else ''}, shape mismatch {bad[:4]}")
            self.load_state_dict({k: sd[k] for k in own if k.startswith("features.")}, strict=False)
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        self.register_buffer("norm_scale", torch.tensor([1.0 / s for s in STD]))
        self.register_buffer("norm_shift", torch.tensor([-m / s for m, s in zip(MEAN, STD)]))
        self.register_buffer("inv_std", torch.tensor([1.0 / s for s in STD]))
        self._packed = {}
        self._ws = {}
        self.to(self.device)

    def packed(self, mode):
        convs = [(i, self.features[i].weight) for i, k, _, _ in self.plan if k == "conv"]
        key = (mode, tuple((w.data_ptr(), w._version) for _, w in convs))
        if self._packed.get("key" + str(mode)) != key:
            self._packed["key" + str(mode)] = key
            self._packed[mode] = {i: ops.pack_conv(w, mode) for i, w in convs}
        return self._packed[mode]

    def forward(self, x, gt):
        if not x.is_cuda:
            raise _abi.HipPathError("ContentLossVGG: the HIP path needs ROCm device tensors (no CPU fallback)")
        return _VggFn.apply(x, gt, self, torch.is_grad_enabled())

    def __repr__(self):
        return "ContentLoss()"

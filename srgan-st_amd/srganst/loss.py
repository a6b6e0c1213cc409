"""Generator criterions on the HIP path.  Mirrors reference loss.py (criterion(sr, gt) -> 0-dim).

StructureTensorLoss  <- reference loss.py:380-413 (+ utils.py:194-280), kernels csrc/st_loss.hip
ContentLossVGG       <- reference loss.py:11-70, in vgg_loss.py (re-exported here)
BestBuddyLoss / GramLoss / PatchwiseStructureTensorLoss <- reference loss.py:78-228, 292-375, kernels csrc/bb_loss.hip
ContentLossDiscriminator <- reference loss.py:231-289, in disc_loss.py (re-exported here)
"""
from __future__ import annotations

import ctypes
import os

import torch
from torch import nn

from . import _abi


class _StLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt, sigma, rho, normalize, ws):
        if x.dtype != torch.float32 or gt.dtype != torch.float32:
            raise _abi.HipPathError("StructureTensorLoss: fp32 only")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape != gt.shape:
            raise _abi.HipPathError(f"StructureTensorLoss: expected [B,3,H,W] pairs, got {tuple(x.shape)} / {tuple(gt.shape)}")
        x = x.contiguous()
        gt = gt.contiguous()
        B, _, H, W = x.shape
        lib = _abi.lib()
        n = ctypes.c_int64()
        _abi.check(lib.sst_st_loss_workspace(B, H, W, ctypes.byref(n)), "sst_st_loss_workspace")
        key = (x.device, B, H, W)
        if ws.get("key") != key:
            ws["key"] = key
            ws["partials"] = torch.empty(n.value, device=x.device, dtype=torch.float32)
            ws["counter"] = torch.zeros(1, device=x.device, dtype=torch.int32)
        gS = torch.empty_like(x)
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        from . import ops
        args = (_abi.ptr(x), _abi.ptr(gt), _abi.ptr(loss), _abi.ptr(gS), _abi.ptr(ws["partials"]), _abi.ptr(ws["counter"]), B, H, W,
                sigma, rho, int(normalize))
        _abi.check(lib.sst_st_loss_fwd(*args, _abi.stream_ptr()), "sst_st_loss_fwd")
        # algorithmic bytes: forward reads sr + gt, backward writes d(sr): 3 x 3*H*W*4 B per image in all
        ops._trace_hbm("st_loss_fwd_kernel", 2.0 * x.numel() * 4, lambda: lib.sst_st_loss_fwd(*args, _abi.stream_ptr()), x, gt, loss, gS, ws)
        ctx.save_for_backward(x, gS)
        ctx.cfg = (sigma, rho)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x, gS = ctx.saved_tensors
        sigma, rho = ctx.cfg
        B, _, H, W = x.shape
        dx = torch.empty_like(x)
        g = grad_out.contiguous().to(torch.float32)
        _abi.check(_abi.lib().sst_st_loss_bwd(_abi.ptr(x), _abi.ptr(gS), _abi.ptr(dx), _abi.ptr(g), 1.0, 0, B, H, W,
                                              sigma, rho, _abi.stream_ptr()), "sst_st_loss_bwd")
        return dx, None, None, None, None, None


class StructureTensorLoss(nn.Module):
    """Same constructor and call protocol as reference loss.py:380-413."""

    def __init__(self, sigma: float = 0.5, rho: float = 2.0, normalize: bool = True):
        super().__init__()
        self.sigma = sigma
        self.rho = rho
        self.normalize = normalize
        self._ws = {}

    def forward(self, x, gt):
        return _StLossFn.apply(x, gt, float(self.sigma), float(self.rho), bool(self.normalize), self._ws)

    def __repr__(self):
        return f"StructureTensorLoss(sigma={self.sigma}, rho={self.rho}, normalize={self.normalize})"


# ------------------------------------------------------------------------------------------------
class _PixelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt, mode, ws):
        from . import ops
        if x.shape != gt.shape or x.dtype != torch.float32:
            raise _abi.HipPathError("pixel loss: fp32 tensors of equal shape")
        x, gt = x.contiguous(), gt.contiguous()
        ctx.save_for_backward(x, gt)
        ctx.mode = mode
        return ops.pixel_loss_fwd(x, gt, mode, ws)

    @staticmethod
    def backward(ctx, grad_out):
        from . import ops
        x, gt = ctx.saved_tensors
        return ops.pixel_loss_bwd(x, gt, ctx.mode, scale_dev=grad_out.contiguous()), None, None, None


class MSELoss(nn.Module):
    """nn.MSELoss() of reference config.py:88-90 on the HIP path."""

    def __init__(self):
        super().__init__()
        self._ws = {}

    def forward(self, x, gt):
        return _PixelLossFn.apply(x, gt, 0, self._ws)


class L1Loss(nn.Module):
    """Pixel-L1 variant named by BASELINE.json configs[0] (one-line config change in the reference)."""

    def __init__(self):
        super().__init__()
        self._ws = {}

    def forward(self, x, gt):
        return _PixelLossFn.apply(x, gt, 1, self._ws)


# ------------------------------------------------------------------------------------------------
FUSE_PIXEL_INTO_ST = os.environ.get("SST_FUSE_PIXEL_ST", "1") != "0"


def _st_pixel_fwd(x, gt, sigma, rho, normalize, ws, mode):
    """Structure-tensor loss and pixel criterion (mode 0 MSE / 1 L1) of (x, gt) in one launch -> (st_loss, pixel_loss, gS)."""
    from . import ops
    if x.dtype != torch.float32 or gt.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or x.shape != gt.shape:
        raise _abi.HipPathError(f"StructureTensorLoss: expected fp32 [B,3,H,W] pairs, got {tuple(x.shape)} / {tuple(gt.shape)}")
    B, _, H, W = x.shape
    lib = _abi.lib()
    n = ctypes.c_int64()
    _abi.check(lib.sst_st_loss_workspace(B, H, W, ctypes.byref(n)), "sst_st_loss_workspace")
    key = (x.device, B, H, W)
    if ws.get("key") != key:
        ws["key"] = key
        ws["partials"] = torch.empty(n.value, device=x.device, dtype=torch.float32)
        ws["counter"] = torch.zeros(1, device=x.device, dtype=torch.int32)
    if ws.get("pix_partials") is None or ws["pix_partials"].numel() != n.value or ws["pix_partials"].device != x.device:
        ws["pix_partials"] = torch.empty(n.value, device=x.device, dtype=torch.float32)
    gS = torch.empty_like(x)
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    pix = torch.empty((), device=x.device, dtype=torch.float32)
    args = (_abi.ptr(x), _abi.ptr(gt), _abi.ptr(loss), _abi.ptr(gS), _abi.ptr(ws["partials"]), _abi.ptr(ws["counter"]), _abi.ptr(pix),
            _abi.ptr(ws["pix_partials"]), int(mode), B, H, W, sigma, rho, int(normalize))
    _abi.check(lib.sst_st_pixel_loss_fwd(*args, _abi.stream_ptr()), "sst_st_pixel_loss_fwd")
    ops._trace_hbm("st_loss_fwd_kernel", 2.0 * x.numel() * 4, lambda: lib.sst_st_pixel_loss_fwd(*args, _abi.stream_ptr()), x, gt, loss, gS, ws, pix)
    return loss, pix, gS


class _CriterionSumFn(torch.autograd.Function):
    """total = sum_i w_i * criterion_i(sr, gt) for HIP-path criterions, as ONE autograd node: the weighted values and the
    total come out of one tiny kernel, and the backward kernels of the terms accumulate into one d(sr) buffer with the
    weight folded into their scale - no per-term mul / add launches (reference train.py:129-140 / warmup.py:79-86 loop)."""

    @staticmethod
    def forward(ctx, sr, gt, terms, weights, ws):
        from . import ops
        if sr.dtype != torch.float32 or sr.shape != gt.shape:
            raise _abi.HipPathError("criterion sum: fp32 tensors of equal shape")
        sr, gt = sr.contiguous(), gt.contiguous()
        lib = _abi.lib()
        raw, saved = [], []
        # one structure-tensor term + one pixel term (the step's usual pair): the pixel criterion rides along in the structure-tensor
        # kernels - same reads of sr / gt, two launches less per step (forward and backward)
        st_i = [i for i, t in enumerate(terms) if isinstance(t, StructureTensorLoss)]
        px_i = [i for i, t in enumerate(terms) if isinstance(t, (MSELoss, L1Loss))]
        ctx.fused_pair = None
        if FUSE_PIXEL_INTO_ST and len(st_i) == 1 and len(px_i) == 1 and sr.dim() == 4 and sr.shape[1] == 3:
            t = terms[st_i[0]]
            mode = 0 if isinstance(terms[px_i[0]], MSELoss) else 1
            st_loss, pix_loss, gS = _st_pixel_fwd(sr, gt, float(t.sigma), float(t.rho), bool(t.normalize), t._ws, mode)
            raw, saved = [None] * len(terms), [None] * len(terms)
            raw[st_i[0]], raw[px_i[0]], saved[st_i[0]] = st_loss, pix_loss, gS
            ctx.fused_pair = (st_i[0], px_i[0], mode)
            for i, t2 in enumerate(terms):
                if raw[i] is None:
                    raise _abi.HipPathError("criterion sum: unexpected term next to the fused pixel + structure-tensor pair")
        else:
            for i, t in enumerate(terms):
                if isinstance(t, StructureTensorLoss):
                    raw.append(_StLossFn.forward(_Ctx(saved), sr, gt, float(t.sigma), float(t.rho), bool(t.normalize), t._ws))
                else:
                    raw.append(ops.pixel_loss_fwd(sr, gt, 0 if isinstance(t, MSELoss) else 1, t._ws))
                    saved.append(None)
        n = len(terms)
        total = torch.empty((), device=sr.device, dtype=torch.float32)
        weighted = torch.empty(n, device=sr.device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * n)(*[_abi.ptr(r) for r in raw])
        wts = (ctypes.c_float * n)(*[float(w) for w in weights])
        _abi.check(lib.sst_weighted_sum(ptrs, wts, n, _abi.ptr(total), _abi.ptr(weighted), _abi.stream_ptr()), "sst_weighted_sum")
        ctx.terms, ctx.weights, ctx.gS = terms, [float(w) for w in weights], saved
        ctx.save_for_backward(sr, gt)
        ctx.mark_non_differentiable(weighted)
        ctx.set_materialize_grads(False)                 # no zeros kernel for the gradient slot of `weighted`
        return total, weighted

    @staticmethod
    def backward(ctx, grad_total, _grad_weighted):
        from . import ops
        sr, gt = ctx.saved_tensors
        g = grad_total.contiguous().to(torch.float32)
        B, _, H, W = sr.shape
        dsr = torch.empty_like(sr)
        if ctx.fused_pair is not None:
            si, pi, mode = ctx.fused_pair
            t = ctx.terms[si]
            bargs = (_abi.ptr(sr), _abi.ptr(gt), _abi.ptr(ctx.gS[si]), _abi.ptr(dsr), _abi.ptr(g), ctx.weights[si], ctx.weights[pi], mode, 0,
                     B, H, W, float(t.sigma), float(t.rho))
            _abi.check(_abi.lib().sst_st_pixel_loss_bwd(*bargs, _abi.stream_ptr()), "sst_st_pixel_loss_bwd")
            ops._trace_hbm("st_loss_bwd_kernel", 1.0 * sr.numel() * 4, lambda: _abi.lib().sst_st_pixel_loss_bwd(*bargs, _abi.stream_ptr()),
                           sr, gt, ctx.gS[si], dsr, g)
            return dsr, None, None, None, None
        for i, (t, w) in enumerate(zip(ctx.terms, ctx.weights)):
            if isinstance(t, StructureTensorLoss):
                bargs = (_abi.ptr(sr), _abi.ptr(ctx.gS[i]), _abi.ptr(dsr), _abi.ptr(g), w, int(i > 0), B, H, W, float(t.sigma), float(t.rho))
                _abi.check(_abi.lib().sst_st_loss_bwd(*bargs, _abi.stream_ptr()), "sst_st_loss_bwd")
                ops._trace_hbm("st_loss_bwd_kernel", 1.0 * sr.numel() * 4, lambda bargs=bargs: _abi.lib().sst_st_loss_bwd(*bargs, _abi.stream_ptr()),
                               sr, ctx.gS[i], dsr, g)
            else:
                ops.pixel_loss_bwd(sr, gt, 0 if isinstance(t, MSELoss) else 1, scale_dev=g, scale_host=w, out=dsr, accumulate=i > 0)
        return dsr, None, None, None, None


class _Ctx:
    """Stand-in for an autograd ctx when a Function's forward is re-used inside another node: keeps the gS tensor."""

    def __init__(self, saved):
        self._saved = saved

    def save_for_backward(self, x, gS):
        self._saved.append(gS)


def fusable(criterion) -> bool:
    return isinstance(criterion, (MSELoss, L1Loss, StructureTensorLoss))


def criterion_sum(sr, gt, terms, weights, ws=None):
    """-> (total, weighted values [len(terms)]) for fusable() criterions of (sr, gt)."""
    return _CriterionSumFn.apply(sr, gt, list(terms), list(weights), ws)


# ------------------------------------------------------------------------------------------------
def _torch_bicubic_taps(in_size: int, out_size: int, device, scale=None):
    """Tap tables of F.interpolate(mode='bicubic', align_corners=False) (A = -0.75, 4 taps, border indices clamped):
    -> (weights [out, 4] fp32, indices [out, 4] int32).  scale: source pixels per output pixel; F.interpolate(scale_factor=f) maps
    coordinates with 1 / f, not with in_size / out_size (they differ when in_size * f is not an integer)."""
    A = -0.75
    o = torch.arange(out_size, dtype=torch.float64)
    src = (o + 0.5) * (float(scale) if scale is not None else in_size / out_size) - 0.5
    x0 = torch.floor(src)
    t = src - x0

    def cc1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def cc2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    w = torch.stack([cc2(t + 1), cc1(t), cc1(1 - t), cc2(2 - t)], dim=1).to(torch.float32)
    idx = (x0.unsqueeze(1) + torch.arange(-1, 3, dtype=torch.float64).unsqueeze(0)).clamp(0, in_size - 1).to(torch.int32)
    return w.contiguous().to(device), idx.contiguous().to(device)


class _BestBuddyGeneralFn(torch.autograd.Function):
    """BestBuddyLoss with any patch geometry (ksize <= 6, pad, stride; reference loss.py:86,116-134) and either matching distance
    (utils.py:157-191): patch tables in global memory (sst_bbg_unfold), tiled matching (sst_bbg_match), the unfold's adjoint for the
    gradient (sst_bbg_fold) - csrc/bb_loss.hip."""

    @staticmethod
    def forward(ctx, x, gt, alpha, beta, l2, ksize, pad, stride, dist_l1, cache):
        if x.dtype != torch.float32 or x.shape != gt.shape or x.dim() != 4 or x.shape[1] != 3:
            raise _abi.HipPathError("BestBuddyLoss: fp32 [B,3,H,W] pairs")
        B, _, H, W = x.shape
        x, gt = x.contiguous(), gt.contiguous()
        lib, dev = _abi.lib(), x.device
        sizes = [(H // s, W // s) for s in (1, 2, 4)]                 # F.interpolate(scale_factor=0.5 / 0.25): floor
        nps = [lib.sst_bbg_patches(h, w, ksize, pad, stride) for h, w in sizes]
        if min(nps) <= 0:
            raise _abi.HipPathError(f"BestBuddyLoss: a {ksize}x{ksize} patch (pad {pad}) does not fit the 1/4-resolution image")
        key = (dev, H, W)
        if cache.get("key") != key:
            cache["key"] = key
            cache["taps"] = [(_torch_bicubic_taps(H, H // s, dev, s), _torch_bicubic_taps(W, W // s, dev, s)) for s in (2, 4)]
        D, ncand, np_ = 3 * ksize * ksize, sum(nps), nps[0]
        cand = torch.empty(B, ncand, D, device=dev, dtype=torch.float32)
        cnrm = torch.empty(B, ncand, device=dev, dtype=torch.float32)
        st = _abi.stream_ptr()
        _abi.check(lib.sst_bbg_unfold(_abi.ptr(gt), _abi.ptr(cand), _abi.ptr(cnrm), B, H, W, ksize, pad, stride, ncand, 0, st), "sst_bbg_unfold")
        off = nps[0]
        for k, s in enumerate((2, 4)):
            (wy, iy), (wx, ix) = cache["taps"][k]
            small = torch.empty(B, 3, H // s, W // s, device=dev, dtype=torch.float32)
            _abi.check(lib.sst_bicubic(_abi.ptr(gt), _abi.ptr(small), _abi.ptr(wy), _abi.ptr(iy), _abi.ptr(wx), _abi.ptr(ix), B * 3, H, W,
                                       H // s, W // s, 4, 4, 0, st), "sst_bicubic")
            _abi.check(lib.sst_bbg_unfold(_abi.ptr(small), _abi.ptr(cand), _abi.ptr(cnrm), B, H // s, W // s, ksize, pad, stride, ncand, off, st),
                       "sst_bbg_unfold")
            off += nps[k + 1]
        srf = torch.empty(B, np_, D, device=dev, dtype=torch.float32)
        _abi.check(lib.sst_bbg_unfold(_abi.ptr(x), _abi.ptr(srf), None, B, H, W, ksize, pad, stride, np_, 0, st), "sst_bbg_unfold")
        ind = torch.empty(B, np_, device=dev, dtype=torch.int32)
        gfeat = torch.empty(B, np_, D, device=dev, dtype=torch.float32)
        partials = torch.empty(lib.sst_bbg_blocks(B, np_), device=dev, dtype=torch.float32)
        _abi.check(lib.sst_bbg_match(_abi.ptr(srf), _abi.ptr(cand), _abi.ptr(cnrm), _abi.ptr(ind), _abi.ptr(gfeat), _abi.ptr(partials), B, np_, ncand,
                                     D, float(alpha), float(beta), int(l2), int(dist_l1), st), "sst_bbg_match")
        dsr = torch.empty_like(x)
        _abi.check(lib.sst_bbg_fold(_abi.ptr(gfeat), _abi.ptr(dsr), B, H, W, ksize, pad, stride, st), "sst_bbg_fold")
        ctx.save_for_backward(dsr)
        ctx.mark_non_differentiable(ind)
        return partials.sum(), ind

    @staticmethod
    def backward(ctx, grad_out, _grad_ind):
        (dsr,) = ctx.saved_tensors
        return dsr * grad_out, None, None, None, None, None, None, None, None, None


class _BestBuddyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt, alpha, beta, l2, cache, gram=0, st_mats=None, dist_l1=0):
        if x.dtype != torch.float32 or x.shape != gt.shape or x.dim() != 4 or x.shape[1] != 3:
            raise _abi.HipPathError("BestBuddyLoss / GramLoss: fp32 [B,3,H,W] pairs")
        B, _, H, W = x.shape
        if H % 12 or W % 12:
            raise _abi.HipPathError("BestBuddyLoss: H and W must be multiples of 12 (3x3 patches at scales 1, 1/2, 1/4)")
        x, gt = x.contiguous(), gt.contiguous()
        lib, dev = _abi.lib(), x.device
        key = (dev, H, W)
        if cache.get("key") != key:
            cache["key"] = key
            cache["taps"] = [(_torch_bicubic_taps(H, H // s, dev), _torch_bicubic_taps(W, W // s, dev)) for s in (2, 4)]
        nps = [(H // s // 3) * (W // s // 3) for s in (1, 2, 4)]
        ncand = sum(nps)
        cand = torch.empty(B, ncand, lib.sst_bb_feature_dim(int(gram)), device=dev, dtype=torch.float32)
        cnrm = torch.empty(B, ncand, device=dev, dtype=torch.float32)
        st = _abi.stream_ptr()
        _abi.check(lib.sst_bb_patches(_abi.ptr(gt), _abi.ptr(cand), _abi.ptr(cnrm), B, H, W, ncand, 0, int(gram), _abi.ptr(st_mats), st), "sst_bb_patches")
        off = nps[0]
        for k, s in enumerate((2, 4)):                       # GT at 1/2 and 1/4 resolution (torch bicubic), then its patches
            (wy, iy), (wx, ix) = cache["taps"][k]
            small = torch.empty(B, 3, H // s, W // s, device=dev, dtype=torch.float32)
            _abi.check(lib.sst_bicubic(_abi.ptr(gt), _abi.ptr(small), _abi.ptr(wy), _abi.ptr(iy), _abi.ptr(wx), _abi.ptr(ix), B * 3, H, W,
                                       H // s, W // s, 4, 4, 0, st), "sst_bicubic")
            _abi.check(lib.sst_bb_patches(_abi.ptr(small), _abi.ptr(cand), _abi.ptr(cnrm), B, H // s, W // s, ncand, off, int(gram),
                                          _abi.ptr(st_mats), st), "sst_bb_patches")
            off += nps[k + 1]
        ind = torch.empty(B, nps[0], device=dev, dtype=torch.int32)
        dsr = torch.empty_like(x)
        partials = torch.empty(lib.sst_bb_blocks(B, H, W), device=dev, dtype=torch.float32)
        _abi.check(lib.sst_bb_match_dist(_abi.ptr(x), _abi.ptr(cand), _abi.ptr(cnrm), _abi.ptr(ind), _abi.ptr(dsr), _abi.ptr(partials), B, H, W,
                                         ncand, float(alpha), float(beta), int(l2), int(gram), _abi.ptr(st_mats), int(dist_l1), st), "sst_bb_match")
        ctx.save_for_backward(dsr)
        ctx.mark_non_differentiable(ind)
        return partials.sum(), ind

    @staticmethod
    def backward(ctx, grad_out, _grad_ind):
        (dsr,) = ctx.saved_tensors
        return dsr * grad_out, None, None, None, None, None, None, None, None


def _dist_l1(dist_norm: str) -> int:
    if dist_norm not in ("l1", "l2"):
        raise NotImplementedError("%s norm has not been supported." % dist_norm)          # utils.py:189
    return int(dist_norm == "l1")


class BestBuddyLoss(nn.Module):
    """Reference loss.py:78-142 (Best-Buddy GAN loss) on the HIP path.  Same constructor: any ksize (<= 6) / pad / stride and
    dist_norm 'l1' or 'l2', criterion 'l1' or 'l2' / 'mse'.  The reference's default geometry (ksize 3, pad 0, stride 3, images with
    H, W multiples of 12) runs the register-resident kernels, everything else the table-based ones (_BestBuddyGeneralFn)."""

    def __init__(self, alpha: float = 1.0, beta: float = 1.0, ksize: int = 3, pad: int = 0, stride: int = 3, dist_norm: str = "l2",
                 criterion: str = "l1") -> None:
        super().__init__()
        if not (1 <= int(ksize) <= 6 and int(pad) >= 0 and int(stride) >= 1):
            raise NotImplementedError("BestBuddyLoss on the HIP path: 1 <= ksize <= 6, pad >= 0, stride >= 1")
        self._dl1 = _dist_l1(dist_norm)
        if criterion not in ("l1", "l2", "mse"):
            raise NotImplementedError("%s criterion has not been implmented." % criterion)
        self.alpha, self.beta, self.ksize, self.pad, self.stride, self.dist_norm = alpha, beta, ksize, pad, stride, dist_norm
        self.l2 = criterion != "l1"
        self._cache = {}
        self.last_index = None          # [B, n_patches] int32: the selected candidate per SR patch (diagnostics / tests)

    def forward(self, x, gt):
        default = (self.ksize, self.pad, self.stride) == (3, 0, 3) and x.shape[2] % 12 == 0 and x.shape[3] % 12 == 0
        if default:
            loss, ind = _BestBuddyFn.apply(x, gt, float(self.alpha), float(self.beta), self.l2, self._cache, 0, None, self._dl1)
        else:
            loss, ind = _BestBuddyGeneralFn.apply(x, gt, float(self.alpha), float(self.beta), self.l2, int(self.ksize), int(self.pad),
                                                  int(self.stride), self._dl1, self._cache)
        self.last_index = ind
        return loss


class GramLoss(nn.Module):
    """Reference loss.py:145-228 (best-buddy matching on the 3x3 gram matrix of every 3x3 patch) on the HIP path.  Same
    constructor; supported: ksize 3 (the reference's reshape to 9 features assumes it, loss.py:200), dist_norm 'l1' or 'l2',
    criterion 'l1' or 'l2' / 'mse'."""

    def __init__(self, alpha: float = 1.0, beta: float = 1.0, ksize: int = 3, dist_norm: str = "l2", criterion: str = "l1") -> None:
        super().__init__()
        if ksize != 3:
            raise NotImplementedError("GramLoss on the HIP path: ksize=3 only")
        self._dl1 = _dist_l1(dist_norm)
        if criterion not in ("l1", "l2", "mse"):
            raise NotImplementedError("%s criterion has not been implmented." % criterion)
        self.alpha, self.beta, self.ksize, self.dist_norm = alpha, beta, ksize, dist_norm
        self.l2 = criterion != "l1"
        self._cache = {}
        self.last_index = None

    def forward(self, x, gt):
        loss, ind = _BestBuddyFn.apply(x, gt, float(self.alpha), float(self.beta), self.l2, self._cache, 1, None, self._dl1)
        self.last_index = ind
        return loss


def _patch_st_matrices(sigma: float, rho: float, device):
    """The three 9x9 linear maps of utils.py:212-233 on a zero-padded 3x3 image, row-major [out pixel][in pixel]:
    Ix = Ax g (derivative taps along H, Gaussian along W), Iy = Ay g, J = K (product)  -> device tensor [3*81]."""
    def taps(s, also_dg=False):                         # utils.py:194-208 (fp32 taps from an int64 arange)
        radius = max(int(4 * s + 0.5), 1)
        xs = torch.arange(-radius, radius + 1)
        s2 = s * s + 1e-12
        phi = torch.exp(-0.5 / s2 * xs ** 2)
        phi = phi / phi.sum()
        return (phi, phi * -xs / s2, radius) if also_dg else (phi, radius)

    g, dg, r1 = taps(sigma, True)
    k, r2 = taps(rho)

    def mat(kv, kh, r):                                  # out (y,x) <- in (y',x'):  kv[y'-y+r] * kh[x'-x+r]
        m = torch.zeros(9, 9)
        for y in range(3):
            for x in range(3):
                for yy in range(3):
                    for xx in range(3):
                        a, b = yy - y + r, xx - x + r
                        if 0 <= a <= 2 * r and 0 <= b <= 2 * r:
                            m[y * 3 + x, yy * 3 + xx] = kv[a] * kh[b]
        return m

    return torch.cat([mat(dg, g, r1).reshape(-1), mat(g, dg, r1).reshape(-1), mat(k, k, r2).reshape(-1)]).to(torch.float32).to(device)


class PatchwiseStructureTensorLoss(nn.Module):
    """Reference loss.py:292-375 (best-buddy matching on the normalised structure tensor of every 3x3 patch) on the HIP path.
    Same constructor; supported: ksize 3, dist_norm 'l1' or 'l2', criterion 'l1' or 'l2' / 'mse'."""

    def __init__(self, sigma: float = 0.5, rho: float = 2, alpha: float = 1.0, beta: float = 1.0, ksize: int = 3, dist_norm: str = "l2",
                 criterion: str = "l1"):
        super().__init__()
        if ksize != 3:
            raise NotImplementedError("PatchwiseStructureTensorLoss on the HIP path: ksize=3 only")
        self._dl1 = _dist_l1(dist_norm)
        if criterion not in ("l1", "l2", "mse"):
            raise NotImplementedError("%s criterion has not been supported." % criterion)
        self.sigma, self.rho, self.alpha, self.beta, self.ksize, self.dist_norm = sigma, rho, alpha, beta, ksize, dist_norm
        self.l2 = criterion != "l1"
        self._cache = {}
        self._mats = None
        self.last_index = None

    def forward(self, x, gt):
        if self._mats is None or self._mats.device != x.device:
            self._mats = _patch_st_matrices(float(self.sigma), float(self.rho), x.device)
        loss, ind = _BestBuddyFn.apply(x, gt, float(self.alpha), float(self.beta), self.l2, self._cache, 2, self._mats, self._dl1)
        self.last_index = ind
        return loss


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_value):
        from . import ops
        logits = logits.contiguous()
        ctx.save_for_backward(logits)
        ctx.t = target_value
        loss, _ = ops.bce_logits(logits, target_value, want_loss=True)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        from . import ops
        (logits,) = ctx.saved_tensors
        _, dl = ops.bce_logits(logits, ctx.t, want_loss=False, want_grad=True, scale_dev=grad_out.contiguous())
        return dl, None


class _AdvTermFn(torch.autograd.Function):
    """total + w * BCEWithLogits(logits, t) -> (new total, w * BCE): the generator's adversarial term (reference train.py:136-140:
    `loss = criterion(D(sr), real) * weight; total += loss`) as one autograd node - one tiny launch for scaling + sum in the forward,
    and the weight folded into the scale of the BCE gradient in the backward (no mul / add launches of autograd)."""

    @staticmethod
    def forward(ctx, total, logits, t, w):
        from . import ops
        logits = logits.contiguous()
        raw, _ = ops.bce_logits(logits, t, want_loss=True)
        out = torch.empty((), device=logits.device, dtype=torch.float32)
        weighted = torch.empty(2, device=logits.device, dtype=torch.float32)
        tot = total.contiguous()
        ptrs = (ctypes.c_void_p * 2)(_abi.ptr(tot), _abi.ptr(raw))
        wts = (ctypes.c_float * 2)(1.0, float(w))
        _abi.check(_abi.lib().sst_weighted_sum(ptrs, wts, 2, _abi.ptr(out), _abi.ptr(weighted), _abi.stream_ptr()), "sst_weighted_sum")
        ctx.save_for_backward(logits)
        ctx.t, ctx.w = t, float(w)
        ctx.mark_non_differentiable(weighted)
        ctx.set_materialize_grads(False)
        return out, weighted

    @staticmethod
    def backward(ctx, g, _gw):
        from . import ops
        (logits,) = ctx.saved_tensors
        _, dl = ops.bce_logits(logits, ctx.t, want_loss=False, want_grad=True, scale_dev=g.contiguous(), scale_host=ctx.w)
        return g, dl, None, None


def adversarial_term(total, logits, crit, label, weight):
    """-> (total + weight * crit(logits, label), weight * crit(...) detached) for crit = BCEWithLogitsLoss (see _AdvTermFn)."""
    out, weighted = _AdvTermFn.apply(total, logits, crit.label_value(label), weight)
    return out, weighted[1]


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() of reference config.py:71-73 / train.py:59.  The reference always calls it
    with a constant-filled label tensor (train.py:113-114: 0.9 or 0); the label value is read once on the
    host when the label tensor is first seen (cached by identity), so there is no per-step sync."""

    def __init__(self):
        super().__init__()
        self._label_cache = {}

    def forward(self, logits, label):
        return _BceFn.apply(logits, self.label_value(label))

    def label_value(self, label):
        if isinstance(label, (int, float)):
            t = float(label)
        else:
            key = (label.data_ptr(), label._version, tuple(label.shape))
            t = self._label_cache.get(key)
            if t is None:
                t = float(label.flatten()[0].item())
                if not bool((label == t).all().item()):
                    raise _abi.HipPathError("BCEWithLogitsLoss (HIP path): label tensor must be constant-filled "
                                            "(reference train.py:113-114 uses full([B,1], 0.9) / zeros)")
                self._label_cache = {key: t}
        return t


from .vgg_loss import ContentLossVGG  # noqa: E402,F401  (reference loss.py:11)
from .disc_loss import ContentLossDiscriminator  # noqa: E402,F401  (reference loss.py:231)

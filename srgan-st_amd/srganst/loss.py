"""Generator criterions on the HIP path.  Mirrors reference loss.py (criterion(sr, gt) -> 0-dim).

StructureTensorLoss  <- reference loss.py:380-413 (+ utils.py:194-280), kernels csrc/st_loss.hip
"""
from __future__ import annotations

import ctypes

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
        _abi.check(lib.sst_st_loss_fwd(_abi.ptr(x), _abi.ptr(gt), _abi.ptr(loss), _abi.ptr(gS), _abi.ptr(ws["partials"]),
                                       _abi.ptr(ws["counter"]), B, H, W, sigma, rho, int(normalize), _abi.stream_ptr()),
                   "sst_st_loss_fwd")
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

"""MATLAB-imresize-compatible antialiased bicubic resampling.  Mirrors reference bicubic.py:15-105
(used by dataset.py:28 to synthesise the LR input: x1/4, 16 taps per axis, weights normalised, border
indices clamped, result rounded to the 1/255 grid and NOT clamped).  Same gather-multiply-sum order
as the reference so results are bit-identical on the CPU.  On a ROCm tensor the resampling is ONE HIP kernel
(csrc/misc.hip: sst_bicubic, SURVEY 8f-2: LR synthesis on device) driven by the same host-built tap tables; the CPU
branch is the data-loader's host code (dataset.py runs in a worker process, like the reference's), not a fallback."""
from __future__ import annotations

import math

import torch
from torch import nn


class NearestNeighbourUpscale(nn.Module):
    def __init__(self, scale_factor: int = 4) -> None:
        super().__init__()
        self.upsampler = nn.Upsample(scale_factor=scale_factor)

    def forward(self, x):
        return self.upsampler(x)


def _cubic(x: torch.Tensor) -> torch.Tensor:
    absx = torch.abs(x)
    absx2 = absx * absx
    absx3 = absx2 * absx
    c1 = (absx <= 1).to(torch.float32)
    c2 = ((1 < absx) & (absx <= 2)).to(torch.float32)
    return (1.5 * absx3 - 2.5 * absx2 + 1) * c1 + (-0.5 * absx3 + 2.5 * absx2 - 4 * absx + 2) * c2


def _contribute(in_size: int, out_size: int, scale: float):
    """bicubic.py:38-81 for one axis -> (weights [out, taps] fp32, indices [out, taps] int64, 0-based)."""
    kernel_width = 4.0 / scale if scale < 1 else 4.0
    x = torch.arange(1, out_size + 1).to(torch.float32)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = torch.floor(u - kernel_width / 2)
    P = int(math.ceil(kernel_width)) + 2
    ind = left.unsqueeze(1) + torch.arange(0, P).to(torch.float32).unsqueeze(0)
    mid = u.unsqueeze(1) - ind
    w = scale * _cubic(mid * scale) if scale < 1 else _cubic(mid)
    w = w / torch.sum(w, 1, keepdim=True)
    ind = torch.clamp(ind, 1, in_size)
    keep = ~torch.eq(w, 0)[0]                    # the reference drops the columns that are zero for the first pixel
    return w[:, keep].contiguous(), (ind[:, keep] - 1).long().contiguous()


class Bicubic(nn.Module):
    def __init__(self, device: str = "cuda:0"):
        super().__init__()
        self.device = device
        self._cache = {}

    def forward(self, input: torch.Tensor, scale: float = 4):
        b, c, h, w = input.shape
        oh, ow = int(h * scale), int(w * scale)
        key = (h, w, scale, input.device)
        if key not in self._cache:
            w0, i0 = _contribute(h, oh, scale)
            w1, i1 = _contribute(w, ow, scale)
            self._cache[key] = tuple(t.to(input.device) for t in (w0, i0, w1, i1))
        w0, i0, w1, i1 = self._cache[key]
        if input.is_cuda:
            from . import _abi
            k2 = key + ("i32",)
            if k2 not in self._cache:
                self._cache[k2] = (i0.to(torch.int32).contiguous(), i1.to(torch.int32).contiguous())
            j0, j1 = self._cache[k2]
            x = input.contiguous().to(torch.float32)
            out = torch.empty(b, c, oh, ow, device=input.device, dtype=torch.float32)
            _abi.check(_abi.lib().sst_bicubic(_abi.ptr(x), _abi.ptr(out), _abi.ptr(w0), _abi.ptr(j0), _abi.ptr(w1), _abi.ptr(j1), b * c,
                                              h, w, oh, ow, w0.shape[1], w1.shape[1], 1, _abi.stream_ptr()), "sst_bicubic")
            return out
        out = input[:, :, i0, :] * w0.unsqueeze(0).unsqueeze(1).unsqueeze(4)        # [b,c,oh,taps,w]
        out = torch.sum(out, dim=3)
        A = out.permute(0, 1, 3, 2)                                                 # [b,c,w,oh]
        out = A[:, :, i1, :] * w1.unsqueeze(0).unsqueeze(1).unsqueeze(4)            # [b,c,ow,taps,oh]
        out = torch.round(255 * torch.sum(out, dim=3).permute(0, 1, 3, 2)) / 255
        return out

"""Generator / Discriminator on the HIP path.  Mirrors reference model.py: same constructors
(config attributes read: model.py:26-28, 89-93), same parameter tree / state-dict keys / init
RNG order (model.py:30-65, 100-136), same forward signatures (model.py:67-71, 138-152):

    Generator(config)(x[B,3,h,w])      -> [B,3,4h,4w] in [0,1]   (NCHW in, NCHW out)
    Discriminator(config)(x[B,3,H,W])  -> [B,1] logits

The nn.Conv2d / BatchNorm2d / PReLU / Linear children are *parameter containers only* (they give
the reference's RNG consumption order, names and shapes); their forward() is never called.  All
arithmetic runs in libsrganst.so through srganst.ops (graph code in gen_graph.py / disc_graph.py).
There is no CPU fallback: calling forward on a CPU tensor raises.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from . import _abi


class _ResidualConvBlock(nn.Module):
    def __init__(self, channels: int) -> None:          # reference model.py:169-178
        super().__init__()
        self.rcb = nn.Sequential(
            nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(channels),
            nn.PReLU(),
            nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(channels),
        )


class _UpsampleBlock(nn.Module):
    def __init__(self, channels: int, upscale_factor: int) -> None:   # reference model.py:155-162
        super().__init__()
        self.upsample_block = nn.Sequential(
            nn.Conv2d(channels, channels * upscale_factor * upscale_factor, (3, 3), (1, 1), (1, 1)),
            nn.PixelShuffle(2),
            nn.PReLU(),
        )


def _require_device(x: Tensor, who: str) -> None:
    if not x.is_cuda:
        raise _abi.HipPathError(f"{who}: the HIP path needs a ROCm device tensor (got {x.device}); there is no CPU "
                                "fallback - the CPU restatement lives in oracle/ and is test infrastructure only")
    if x.dtype != torch.float32:
        raise _abi.HipPathError(f"{who}: fp32 only (got {x.dtype})")


class Generator(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        in_channels: int = config.MODEL.G_IN_CHANNEL
        out_channels: int = config.MODEL.G_OUT_CHANNEL
        channels: int = config.MODEL.G_N_CHANNEL
        num_rcb: int = config.MODEL.G_N_RCB
        upscale: int = config.DATA.UPSCALE_FACTOR
        if channels % 4:
            raise ValueError("G_N_CHANNEL must be a multiple of 4 on the HIP path")
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, channels, (9, 9), (1, 1), (4, 4)), nn.PReLU())
        self.trunk = nn.Sequential(*[_ResidualConvBlock(channels) for _ in range(num_rcb)])
        self.conv2 = nn.Sequential(nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1), bias=False),
                                   nn.BatchNorm2d(channels))
        upsampling = []
        if upscale in (2, 4, 8):
            for _ in range(int(math.log(upscale, 2))):
                upsampling.append(_UpsampleBlock(channels, 2))
        elif upscale == 3:
            # reference model.py:122-123,160 hard-codes PixelShuffle(2) behind a x9-channel conv: shape-inconsistent
            raise NotImplementedError("UPSCALE_FACTOR 3 is broken in the reference (model.py:123,160); x2/x4/x8 only")
        self.upsampling = nn.Sequential(*upsampling)
        self.conv3 = nn.Conv2d(channels, out_channels, (9, 9), (1, 1), (4, 4))
        for module in self.modules():                   # reference model.py:130-136
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
            elif isinstance(module, nn.BatchNorm2d):
                nn.init.constant_(module.weight, 1)
        self._names = [n for n, _ in self.named_parameters()]

    def forward(self, x: Tensor) -> Tensor:
        return self._forward_impl(x)

    def _forward_impl(self, x: Tensor) -> Tensor:
        from . import gen_graph
        _require_device(x, "Generator.forward")
        params = [p for _, p in self.named_parameters()]
        return gen_graph.GeneratorFn.apply(x, self, torch.is_grad_enabled(), *params)


class Discriminator(nn.Module):
    """reference model.py:7-71.  ``config.DATA.GT_IMAGE_SIZE`` sizes the classifier (8*C*(HR/16)^2 inputs):
    identical to the reference at 96 px (model.py:62 hard-codes 6*6), defined for 192 px too (BASELINE configs[4])."""

    def __init__(self, config) -> None:
        super().__init__()
        in_channels = config.MODEL.D_IN_CHANNEL
        channels = config.MODEL.D_N_CHANNEL
        out_channels = config.MODEL.D_OUT_CHANNEL
        image_size = int(config.DATA.get("GT_IMAGE_SIZE", 96)) if hasattr(config, "DATA") else 96
        if image_size % 16:
            raise ValueError("GT_IMAGE_SIZE must be a multiple of 16")
        fs = image_size // 16
        self.features = nn.Sequential(
            nn.Conv2d(in_channels, channels, (3, 3), (1, 1), (1, 1), bias=True),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(channels, channels, (3, 3), (2, 2), (1, 1), bias=False),
            nn.BatchNorm2d(channels),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(channels, int(2 * channels), (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(int(2 * channels)),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(int(2 * channels), int(2 * channels), (3, 3), (2, 2), (1, 1), bias=False),
            nn.BatchNorm2d(int(2 * channels)),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(int(2 * channels), int(4 * channels), (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(int(4 * channels)),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(int(4 * channels), int(4 * channels), (3, 3), (2, 2), (1, 1), bias=False),
            nn.BatchNorm2d(int(4 * channels)),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(int(4 * channels), int(8 * channels), (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(int(8 * channels)),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(int(8 * channels), int(8 * channels), (3, 3), (2, 2), (1, 1), bias=False),
            nn.BatchNorm2d(int(8 * channels)),
            nn.LeakyReLU(0.2, True),
        )
        self.classifier = nn.Sequential(
            nn.Linear(int(8 * channels) * fs * fs, 1024),
            nn.LeakyReLU(0.2, True),
            nn.Linear(1024, out_channels),
        )
        self.image_size = image_size

    def forward(self, x: Tensor) -> Tensor:
        from . import disc_graph
        _require_device(x, "Discriminator.forward")
        params = [p for _, p in self.named_parameters()]
        return disc_graph.DiscriminatorFn.apply(x, self, torch.is_grad_enabled(), *params)

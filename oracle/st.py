"""Oracle: structure-tensor loss (CPU, plain torch).  TEST INFRASTRUCTURE ONLY.

Restates, batched and without ``vmap``:
  * get_gaussian_kernel      reference utils.py:194-208
  * structure_tensor         reference utils.py:212-233
  * normalize                reference utils.py:236-239
  * compute_invS1xS2         reference utils.py:242-254
  * compute_eigenvalues      reference utils.py:257-266
  * compute_distance         reference utils.py:269-280
  * StructureTensorLoss      reference loss.py:380-413 (sigma=.5, rho=2, normalize=True)
  * torchvision Grayscale    ITU-R 601 weights (0.2989, 0.587, 0.114) - torchvision is
                             absent from the image, restated from its documentation.
All functions are dtype-generic (fp32 to mirror the reference, fp64 as "truth").
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

GRAY_WEIGHTS = (0.2989, 0.587, 0.114)


def gaussian_kernel(sigma: float, also_dg: bool = False, radius: int | None = None,
                    dtype=torch.float32):
    """utils.py:194-208.  The reference builds the taps in fp32 from an int64 arange."""
    if radius is None:
        radius = max(int(4 * sigma + 0.5), 1)
    x = torch.arange(-radius, radius + 1)
    sigma2 = (sigma * sigma) + 1e-12
    phi = torch.exp(-0.5 / sigma2 * x ** 2)          # fp32, like the reference
    phi = phi / phi.sum()
    if also_dg:
        return phi.to(dtype), (phi * -x / sigma2).to(dtype)
    return phi.to(dtype)


def grayscale(x: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] -> [B,1,H,W]; torchvision.transforms.Grayscale (loss.py:400-401)."""
    r, g, b = x.unbind(dim=-3)
    return (GRAY_WEIGHTS[0] * r + GRAY_WEIGHTS[1] * g + GRAY_WEIGHTS[2] * b).unsqueeze(-3)


def structure_tensor(im: torch.Tensor, sigma: float = 1.0, rho: float = 10.0) -> torch.Tensor:
    """im [B,1,H,W] -> S [B,3,H,W] = (Jxx, Jyy, Jxy).  utils.py:212-233.

    Every conv2d zero-pads its own input ('same'), and conv2d is cross-correlation.
    """
    g, dg = (t.to(im.device) for t in gaussian_kernel(sigma, also_dg=True, dtype=im.dtype))
    h = (1, 1, -1, 1)
    w = (1, 1, 1, -1)
    Ix = F.conv2d(im, dg.reshape(h), padding="same")
    Ix = F.conv2d(Ix, g.reshape(w), padding="same")
    Iy = F.conv2d(im, g.reshape(h), padding="same")
    Iy = F.conv2d(Iy, dg.reshape(w), padding="same")
    k = gaussian_kernel(rho, dtype=im.dtype).to(im.device)

    def integ(p):
        p = F.conv2d(p, k.reshape(h), padding="same")
        return F.conv2d(p, k.reshape(w), padding="same")

    return torch.cat((integ(Ix * Ix), integ(Iy * Iy), integ(Ix * Iy)), dim=1)


def normalize(S: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils.py:236-239 on [B,3,H,W]."""
    d = S[:, 0] * S[:, 1] - S[:, 2] ** 2
    return S / torch.sqrt(d + eps).unsqueeze(1)


def inv_s1_x_s2(S1: torch.Tensor, S2: torch.Tensor, _normalize: bool = True) -> torch.Tensor:
    """utils.py:242-254 -> [B,4,H,W] = (A, B, C, D)."""
    if _normalize:
        S1 = normalize(S1)
        S2 = normalize(S2)
    A = S1[:, 1] * S2[:, 0] - S1[:, 2] * S2[:, 2]
    B = S1[:, 0] * S2[:, 1] - S1[:, 2] * S2[:, 2]
    C = S1[:, 1] * S2[:, 2] - S1[:, 2] * S2[:, 1]
    D = S1[:, 0] * S2[:, 2] - S1[:, 2] * S2[:, 0]
    return torch.stack((A, B, C, D), dim=1)


def eigenvalues(M: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils.py:257-266 -> [B,2,H,W]."""
    ApB = M[:, 0] + M[:, 1]
    disc = ApB ** 2 - 4 * (M[:, 0] * M[:, 1] - M[:, 2] * M[:, 3])
    disc = torch.clamp(disc, min=0 + eps)
    r = torch.sqrt(disc)
    return torch.stack((0.5 * (ApB - r), 0.5 * (ApB + r)), dim=1)


def distance(L: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils.py:269-280 -> [B,H,W]."""
    L = torch.clamp(L, min=1)
    L = torch.log(L) ** 2
    return torch.sqrt(L.sum(dim=1) + eps)


def st_loss(x: torch.Tensor, gt: torch.Tensor, sigma: float = 0.5, rho: float = 2.0,
            normalize_: bool = True) -> torch.Tensor:
    """StructureTensorLoss.forward (loss.py:399-413): mean over pixels, then over batch."""
    s_x = structure_tensor(grayscale(x), sigma, rho)
    s_gt = structure_tensor(grayscale(gt), sigma, rho)
    d = distance(eigenvalues(inv_s1_x_s2(s_x, s_gt, normalize_)))
    return d.mean(dim=(1, 2)).mean()


def st_loss_and_grad(x: torch.Tensor, gt: torch.Tensor, sigma: float = 0.5, rho: float = 2.0,
                     normalize_: bool = True):
    """Loss and d(loss)/d(x) by autograd - what the HIP fwd+bwd kernels must reproduce."""
    x = x.detach().clone().requires_grad_(True)
    loss = st_loss(x, gt, sigma, rho, normalize_)
    (gx,) = torch.autograd.grad(loss, x)
    return loss.detach(), gx

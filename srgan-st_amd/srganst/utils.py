"""Host-side helpers.  Mirrors reference utils.py:13-154 (seeding, checkpoint key handling, image
metrics) without cv2 / torchvision (not installed; the arithmetic is restated):

    init_random_seed   utils.py:13-22
    load_state_dict    utils.py:25-59   (strips the 10-char ``_orig_mod.`` prefix, drops shape mismatches)
    tensor2img         utils.py:62-87   (squeeze, clamp, RGB->BGR, x255 round)
    PSNR               utils.py:90-102
    SSIM               utils.py:105-129 (11x11 sigma 1.5 Gaussian, valid region [5:-5]; cv2.filter2D restated -
                                         "parity unpinned": cv2 is absent, no reference output exists to pin it)
    bgr2ycbcr          utils.py:132-154

The structure-tensor maths of utils.py:194-280 lives in csrc/st_loss.hip (HIP) and oracle/st.py (checker).
"""
from __future__ import annotations

import math
import random
from collections import OrderedDict

import numpy as np
import torch
from torch import nn


def init_random_seed(seed: int = 0) -> None:
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def load_state_dict(model: nn.Module, state_dict: dict) -> nn.Module:
    model_is_compiled = "_orig_mod" in list(state_dict.keys())[0]
    model_state_dict = model.state_dict()
    new_state_dict = OrderedDict()
    for layer_name, weights in state_dict.items():
        new_state_dict[layer_name[10:] if model_is_compiled else layer_name] = weights
    new_state_dict = {k: v for k, v in new_state_dict.items()
                      if k in model_state_dict.keys() and v.size() == model_state_dict[k].size()}
    model_state_dict.update(new_state_dict)
    model.load_state_dict(model_state_dict)
    return model


def _make_grid(t: torch.Tensor, nrow: int) -> torch.Tensor:
    """torchvision.utils.make_grid(padding=0, normalize=False) for [N,C,H,W]."""
    n, c, h, w = t.shape
    ncol = min(nrow, n)
    nr = int(math.ceil(n / ncol))
    grid = t.new_zeros((c, h * nr, w * ncol))
    for k in range(n):
        r, q = divmod(k, ncol)
        grid[:, r * h:(r + 1) * h, q * w:(q + 1) * w] = t[k]
    return grid


def tensor2img(tensor: torch.Tensor, out_type=np.uint8, min_max=(0, 1)) -> np.ndarray:
    tensor = tensor.squeeze().float().cpu().clamp_(*min_max)
    tensor = (tensor - min_max[0]) / (min_max[1] - min_max[0])
    n_dim = tensor.dim()
    if n_dim == 4:
        img_np = _make_grid(tensor, nrow=int(math.sqrt(len(tensor)))).numpy()
        img_np = np.transpose(img_np[[2, 1, 0], :, :], (1, 2, 0))
    elif n_dim == 3:
        img_np = np.transpose(tensor.numpy()[[2, 1, 0], :, :], (1, 2, 0))
    elif n_dim == 2:
        img_np = tensor.numpy()
    else:
        raise TypeError("Only support 4D, 3D and 2D tensor. But received with dimension: {:d}".format(n_dim))
    if out_type == np.uint8:
        img_np = (img_np * 255.0).round()
    return img_np.astype(out_type)


def PSNR(img1: np.ndarray, img2: np.ndarray) -> float:
    img1 = img1.astype(np.float64)
    img2 = img2.astype(np.float64)
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gaussian_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k /= k.sum()
    return np.outer(k, k)


def _filter_valid(img: np.ndarray, window: np.ndarray) -> np.ndarray:
    """cv2.filter2D(img, -1, window)[5:-5, 5:-5] == 'valid' correlation (the cropped border never reads outside)."""
    kh, kw = window.shape
    H, W = img.shape
    out = np.zeros((H - kh + 1, W - kw + 1), dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            out += window[i, j] * img[i:i + H - kh + 1, j:j + W - kw + 1]
    return out


def SSIM(img1: np.ndarray, img2: np.ndarray) -> float:
    C1 = (0.01 * 255) ** 2
    C2 = (0.03 * 255) ** 2
    img1 = img1.astype(np.float64)
    img2 = img2.astype(np.float64)
    window = _gaussian_window(11, 1.5)
    mu1 = _filter_valid(img1, window)
    mu2 = _filter_valid(img2, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = _filter_valid(img1 ** 2, window) - mu1_sq
    sigma2_sq = _filter_valid(img2 ** 2, window) - mu2_sq
    sigma12 = _filter_valid(img1 * img2, window) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return float(ssim_map.mean())


def bgr2ycbcr(img: np.ndarray, only_y: bool = True) -> np.ndarray:
    in_img_type = img.dtype
    if in_img_type != np.uint8:
        img = img * 255.0                       # the reference scales its argument in place (utils.py:143)
    if only_y:
        rlt = np.dot(img, [24.966, 128.553, 65.481]) / 255.0 + 16.0
    else:
        rlt = np.matmul(img, [[24.966, 112.0, -18.214], [128.553, -74.203, -93.786], [65.481, -37.797, 112.0]]) / 255.0 + [16, 128, 128]
    if in_img_type == np.uint8:
        rlt = rlt.round()
    else:
        rlt /= 255.0
    return rlt.astype(in_img_type)


def start_workers(loader):
    """Fork a DataLoader's persistent workers NOW, with the garbage collector frozen across the fork.

    The workers are forked from a process that owns HIP objects.  If unreachable-but-uncollected Python garbage that
    holds device tensors or hipGraphs exists at fork time (e.g. the step engine of an earlier warmup() call, which sits in
    a reference cycle), a cyclic collection INSIDE a worker would run those objects' destructors against a HIP runtime that
    does not survive fork() - observed as "DataLoader worker killed by signal: Segmentation fault".  Collecting first and
    freezing the survivors (the documented CPython recipe for fork) keeps the children's collector away from them."""
    import gc
    if getattr(loader, "num_workers", 0) == 0 or not getattr(loader, "persistent_workers", False):
        return loader
    gc.collect()
    gc.freeze()
    try:
        iter(loader)          # creates the persistent iterator = forks the workers; later iter() calls re-use them
    finally:
        gc.unfreeze()
    return loader

"""Datasets.  Mirrors reference dataset.py:9-64 (TrainImageDataset: HR crop file -> /255 -> CPU
Bicubic x1/upscale; TestImageDataset: paired GT / LR directories) with PIL instead of
torchvision.io (not installed), plus the synthetic DIV2K-shaped dataset the benchmark uses."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset

from .bicubic import Bicubic


def read_image(path: str) -> Tensor:
    """uint8 [3,H,W] RGB, like torchvision.io.read_image."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()


def absoluteFilePaths(directory):
    for dirpath, _, filenames in os.walk(directory):
        for f in filenames:
            yield os.path.abspath(os.path.join(dirpath, f))


class TrainImageDataset(Dataset):
    def __init__(self, gt_image_dir: str, upscale_factor: int) -> None:
        super().__init__()
        self.image_file_names = [os.path.join(gt_image_dir, n) for n in absoluteFilePaths(gt_image_dir)]
        self.upscale_factor = upscale_factor
        self.bicubic = Bicubic("cpu")

    def __getitem__(self, batch_index: int):
        gt_tensor = read_image(self.image_file_names[batch_index]).float().unsqueeze(0) / 255.0
        lr_tensor = self.bicubic(gt_tensor, scale=1.0 / self.upscale_factor)
        return gt_tensor.squeeze(), lr_tensor.squeeze()

    def __len__(self) -> int:
        return len(self.image_file_names)


class TestImageDataset(Dataset):
    def __init__(self, test_gt_images_dir: str, test_lr_images_dir: str) -> None:
        super().__init__()
        self.gt_image_file_names = sorted(x for x in absoluteFilePaths(test_gt_images_dir) if not os.path.basename(x).startswith("."))
        self.lr_image_file_names = sorted(x for x in absoluteFilePaths(test_lr_images_dir) if not os.path.basename(x).startswith("."))

    def __getitem__(self, batch_index: int):
        gt_tensor = read_image(self.gt_image_file_names[batch_index]).float() / 255.0
        lr_tensor = read_image(self.lr_image_file_names[batch_index]).float() / 255.0
        return gt_tensor, lr_tensor

    def __len__(self) -> int:
        return len(self.gt_image_file_names)


class SyntheticImageDataset(Dataset):
    """DIV2K-shaped synthetic crops on the 1/255 grid: kind 'noise' = uniform u8,
    'lowfreq' = bicubic-upsampled coarse noise.  LR = Bicubic(x1/upscale) like TrainImageDataset."""

    def __init__(self, length: int, hr: int = 96, upscale: int = 4, seed: int = 0, kind: str = "lowfreq"):
        self.length, self.hr, self.upscale, self.seed, self.kind = length, hr, upscale, seed, kind
        self.bicubic = Bicubic("cpu")

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        if self.kind == "noise":
            gt = torch.randint(0, 256, (1, 3, self.hr, self.hr), generator=g, dtype=torch.uint8).float() / 255.0
        else:
            base = torch.rand(1, 3, max(self.hr // 8, 2), max(self.hr // 8, 2), generator=g)
            gt = torch.nn.functional.interpolate(base, size=(self.hr, self.hr), mode="bicubic", align_corners=False)
            gt = torch.round(gt.clamp(0, 1) * 255) / 255
        lr = self.bicubic(gt, scale=1.0 / self.upscale)
        return gt.squeeze(0), lr.squeeze(0)

"""Validation / test path.  Mirrors reference validate.py:18-137: batch-1 full-image generator forward
under no_grad -> tensor2img -> /255 -> Y channel -> PSNR / SSIM (no border shave), mean +- CI.
The generator forward is the HIP path in eval mode (BatchNorm folded to scale/shift from running stats)."""
from __future__ import annotations

import argparse
import os
from statistics import NormalDist

import numpy as np
import torch
from torch.utils.data import DataLoader

from .bicubic import Bicubic, NearestNeighbourUpscale
from .config import Config
from .dataset import TestImageDataset
from .model import Generator
from .utils import PSNR, SSIM, bgr2ycbcr, load_state_dict, tensor2img


def confidence_interval(data, confidence=0.95):
    if len(data) < 2:
        return 0.0
    dist = NormalDist.from_samples(data)
    z = NormalDist().inv_cdf((1 + confidence) / 2.0)
    return dist.stdev * z / ((len(data) - 1) ** 0.5)


def image_metrics(sr: torch.Tensor, hr: torch.Tensor):
    """(psnr, ssim) of one SR/HR pair exactly as validate.py:79-99."""
    output = tensor2img(sr).astype(np.float32) / 255.0
    gt = tensor2img(hr).astype(np.float32) / 255.0
    output = bgr2ycbcr(output, only_y=True)
    gt = bgr2ycbcr(gt, only_y=True)
    return PSNR(output * 255, gt * 255), SSIM(output * 255, gt * 255)


def _save_png(path, bgr_u8):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr_u8[..., ::-1])).save(path)


def _validate(generator, val_loader, config, save_images=False, concat_with_gt=False, save_metrics=False):
    file = None
    if save_metrics:
        path = os.path.join(config.DATA.TEST_SR_IMAGES_DIR, config.EXP.NAME)
        os.makedirs(path, exist_ok=True)
        file = open(os.path.join(path, "_metrics.txt"), mode="w")
    all_psnr, all_ssim = [], []
    with torch.no_grad():
        for idx, (hr_img, lr_img) in enumerate(val_loader):
            lr_img = lr_img.to(config.DEVICE)
            hr_img = hr_img.to(config.DEVICE)
            output = generator(lr_img)
            if save_images:
                path = os.path.join(config.DATA.TEST_SR_IMAGES_DIR, config.EXP.NAME)
                os.makedirs(path, exist_ok=True)
                o, g = tensor2img(output), tensor2img(hr_img)
                _save_png(f"{path}/{idx}.png", np.concatenate([o, g], axis=1) if concat_with_gt else o)
            psnr, ssim = image_metrics(output, hr_img)
            all_psnr.append(psnr)
            all_ssim.append(ssim)
            if file:
                file.write(f"{idx}.png | PSNR: {psnr:.2f} | SSIM: {ssim:.4f}\n")
    avg_psnr = sum(all_psnr) / len(all_psnr)
    avg_ssim = sum(all_ssim) / len(all_ssim)
    out = (f"[Test] | PSNR: {avg_psnr:.2f} ± {confidence_interval(all_psnr):.2f} | "
           f"SSIM: {avg_ssim:.4f} ± {confidence_interval(all_ssim):.4f} | \n")
    print(out)
    if file:
        file.write("\n" + out + "\n")
        file.close()
    return avg_psnr, avg_ssim


def test(config: Config, save_images: bool = True, g_path: str = None, concat_w_gt: bool = False, dataset=None):
    if not g_path:
        g_path = f"results/{config.EXP.NAME}/g_best.pth"
    ds = dataset if dataset is not None else TestImageDataset(config.DATA.TEST_GT_IMAGES_DIR, config.DATA.TEST_LR_IMAGES_DIR)
    loader = DataLoader(ds, batch_size=1, shuffle=False, num_workers=0, drop_last=False)
    if config.EXP.NAME == "bicubic":
        generator = Bicubic(device=config.DEVICE).to(config.DEVICE)
    elif config.EXP.NAME == "nearest":
        generator = NearestNeighbourUpscale(config.DATA.UPSCALE_FACTOR).to(config.DEVICE)
    else:
        generator = Generator(config).to(config.DEVICE)
        generator = load_state_dict(generator, torch.load(g_path, map_location=config.DEVICE, weights_only=True))
        generator.eval()
    return _validate(generator, loader, config, save_images=save_images, concat_with_gt=concat_w_gt, save_metrics=True)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--name", type=str, default=None)
    parser.add_argument("--g-path", type=str, default=None)
    parser.add_argument("--test-set", type=str, default=None)
    parser.add_argument("--no-images", action="store_true")
    a = parser.parse_args()
    cfg = Config()
    if a.name:
        cfg.EXP.NAME = a.name
    if a.test_set:
        cfg.DATA.TEST_SET = a.test_set
        cfg.DATA.TEST_GT_IMAGES_DIR = f"/work3/{cfg.EXP.USER}/data/{a.test_set}/GTmod12"
        cfg.DATA.TEST_LR_IMAGES_DIR = f"/work3/{cfg.EXP.USER}/data/{a.test_set}/LRbicx4"
    test(cfg, save_images=not a.no_images, g_path=a.g_path)

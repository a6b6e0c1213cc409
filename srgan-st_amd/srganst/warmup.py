"""SRResNet pre-training driver.  Mirrors reference warmup.py:14-147: seed, Generator, Adam(eps 1e-4),
loaders, per-epoch step loop, validation, checkpoints (g_last / g_best / g_epochN with the reference's
state-dict keys).  The step itself is srganst.engine.WarmupEngine (HIP kernels, hipGraph, RCCL)."""
from __future__ import annotations

import os

import torch
from torch.utils.data import DataLoader

from . import dist as sdist
from .bicubic import Bicubic
from .config import Config
from .dataset import TestImageDataset, TrainImageDataset
from .engine import WarmupEngine
from .model import Generator
from .utils import init_random_seed, start_workers
from .validate import _validate


class _NullWriter:
    def add_scalar(self, *a, **k): pass
    def add_text(self, *a, **k): pass


def _writer(name):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(f"tensorboard/{name}")
    except Exception:
        return _NullWriter()


def warmup(config: Config, train_dataset=None, test_dataset=None, max_steps_per_epoch=None):
    rank, local, world = sdist.init_from_env(config.DIST.BACKEND)
    init_random_seed(config.DATA.SEED)
    best_psnr = best_ssim = 0.0
    generator = Generator(config).to(config.DEVICE)
    sdist.broadcast_module(generator)
    engine = WarmupEngine(config, generator)
    train_ds = train_dataset or TrainImageDataset(config.DATA.TRAIN_GT_IMAGES_DIR, config.DATA.UPSCALE_FACTOR)
    test_ds = test_dataset or TestImageDataset(config.DATA.TEST_GT_IMAGES_DIR, config.DATA.TEST_LR_IMAGES_DIR)
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds, world, rank, shuffle=True) if world > 1 else None
    train_loader = DataLoader(train_ds, batch_size=config.DATA.BATCH_SIZE, shuffle=sampler is None, sampler=sampler,
                              num_workers=1, pin_memory=True, drop_last=True, persistent_workers=True)
    test_loader = DataLoader(test_ds, batch_size=1, shuffle=False, num_workers=0, drop_last=False)
    device_bicubic = Bicubic(config.DEVICE)
    start_workers(train_loader)                  # fork the loader workers with the collector frozen (see utils.start_workers)
    writer = _writer(config.EXP.NAME) if rank == 0 else _NullWriter()
    writer.add_text("Config/Params", config.get_all_params())
    batches_done = 0
    for epoch in range(config.EXP.START_EPOCH, config.EXP.N_EPOCHS):
        if rank == 0:
            print(f"Beginning warmup epoch: {epoch+1}")
        generator.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch_num, (gt, lr) in enumerate(train_loader):
            if max_steps_per_epoch is not None and batch_num >= max_steps_per_epoch:
                break
            batches_done += 1
            # host batch straight into the engine's static input buffers when they exist and fit (copy_ returns the buffer)
            fits = engine.gt is not None and engine.gt.shape == gt.shape
            gt = engine.gt.copy_(gt, non_blocking=True) if fits else gt.to(device=config.DEVICE, non_blocking=True)
            if config.KERNEL.LR_ON_DEVICE:
                lr = device_bicubic(gt, scale=1.0 / config.DATA.UPSCALE_FACTOR)
            elif fits and engine.lr.shape == lr.shape:
                lr = engine.lr.copy_(lr, non_blocking=True)
            else:
                lr = lr.to(device=config.DEVICE, non_blocking=True)
            loss_values = engine.step(gt, lr)
            if batch_num % config.LOG_TRAIN_PERIOD != 0 or rank != 0:
                continue
            vals = {k: float(v) for k, v in loss_values.items()}          # the only host sync, on log steps
            writer.add_scalar("Train/G_Loss", sum(vals.values()), batches_done)
            for name, v in vals.items():
                writer.add_scalar(f"Train/G_{name}", v, batches_done)
            print(f"[Epoch {epoch+1}/{config.EXP.N_EPOCHS}] [Batch {batch_num}/{len(train_loader)}] [G losses: {vals}]")
        generator.eval()
        if rank == 0:
            psnr, ssim = _validate(generator, test_loader, config)
            if epoch % config.LOG_VALIDATION_PERIOD == 0:
                print(f"[Test: {epoch+1}/{config.EXP.N_EPOCHS}] [PSNR: {psnr}] [SSIM: {ssim}]")
            writer.add_scalar("Test/PSNR", psnr, epoch + 1)
            writer.add_scalar("Test/SSIM", ssim, epoch + 1)
            results_dir = f"results/{config.EXP.NAME}"
            os.makedirs(results_dir, exist_ok=True)
            torch.save(generator.state_dict(), results_dir + "/g_last.pth")
            if best_psnr < psnr and best_ssim < ssim:
                torch.save(generator.state_dict(), results_dir + "/g_best.pth")
                best_psnr, best_ssim = psnr, ssim
            if 0 < epoch and epoch % config.G_CHECKPOINT_INTERVAL == 0:
                torch.save(generator.state_dict(), results_dir + f"/g_epoch{epoch}.pth")
    engine.close()
    return generator


if __name__ == "__main__":
    warmup(Config())

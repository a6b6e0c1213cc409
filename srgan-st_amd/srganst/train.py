"""SRGAN training driver.  Mirrors reference train.py:16-226: seeds, D then G construction, two Adams
(eps 1e-4), MultiStepLR(milestones=[10], gamma), optional warm weights, the G-then-D iteration
(engine.TrainEngine = train.py:116-164), per-epoch validation and checkpoints with reference key names."""
from __future__ import annotations

import os

import torch
from torch.optim import lr_scheduler
from torch.utils.data import DataLoader

from . import dist as sdist
from .bicubic import Bicubic
from .config import Config
from .dataset import TestImageDataset, TrainImageDataset
from .engine import TrainEngine
from .model import Discriminator, Generator
from .utils import init_random_seed, load_state_dict, start_workers
from .validate import _validate
from .warmup import _NullWriter, _writer


def train(config: Config, train_dataset=None, test_dataset=None, max_steps_per_epoch=None):
    rank, local, world = sdist.init_from_env(config.DIST.BACKEND)
    init_random_seed(config.DATA.SEED)
    best_psnr = best_ssim = 0.0
    train_ds = train_dataset or TrainImageDataset(config.DATA.TRAIN_GT_IMAGES_DIR, config.DATA.UPSCALE_FACTOR)
    test_ds = test_dataset or TestImageDataset(config.DATA.TEST_GT_IMAGES_DIR, config.DATA.TEST_LR_IMAGES_DIR)
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds, world, rank, shuffle=True) if world > 1 else None
    train_loader = DataLoader(train_ds, batch_size=config.DATA.BATCH_SIZE, shuffle=sampler is None, sampler=sampler,
                              num_workers=1, pin_memory=True, drop_last=True, persistent_workers=True)
    test_loader = DataLoader(test_ds, batch_size=1, shuffle=False, num_workers=0, drop_last=False)
    device_bicubic = Bicubic(config.DEVICE)
    start_workers(train_loader)                  # fork the loader workers with the collector frozen (see utils.start_workers)
    discriminator = Discriminator(config).to(config.DEVICE)     # train.py:52-53: D is constructed before G
    generator = Generator(config).to(config.DEVICE)
    if config.MODEL.G_CONTINUE_FROM_WARMUP:
        generator = load_state_dict(generator, torch.load(config.MODEL.G_WARMUP_WEIGHTS, map_location=config.DEVICE, weights_only=True))
    if config.MODEL.D_CONTINUE_FROM_WARMUP:
        discriminator = load_state_dict(discriminator, torch.load(config.MODEL.D_WARMUP_WEIGHTS, map_location=config.DEVICE, weights_only=True))
    sdist.broadcast_module(generator)
    sdist.broadcast_module(discriminator)
    engine = TrainEngine(config, generator, discriminator)
    d_scheduler = lr_scheduler.MultiStepLR(engine.d_opt, milestones=[10], gamma=config.SCHEDULER.GAMMA)
    g_scheduler = lr_scheduler.MultiStepLR(engine.g_opt, milestones=[10], gamma=config.SCHEDULER.GAMMA)
    writer = _writer(config.EXP.NAME) if rank == 0 else _NullWriter()
    writer.add_text("Config/Params", config.get_all_params())
    for epoch in range(config.EXP.START_EPOCH, config.EXP.N_EPOCHS):
        if rank == 0:
            print(f"Beginning train epoch: {epoch+1}")
        generator.train()
        discriminator.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        engine.batch_num = 0
        d_loss = None
        for batch_num, (gt, lr) in enumerate(train_loader):
            if max_steps_per_epoch is not None and batch_num >= max_steps_per_epoch:
                break
            # host batch straight into the engine's static input buffers when they exist and fit (copy_ returns the buffer)
            fits = engine.gt is not None and engine.gt.shape == gt.shape
            gt = engine.gt.copy_(gt, non_blocking=True) if fits else gt.to(device=config.DEVICE, non_blocking=True)
            if config.KERNEL.LR_ON_DEVICE:
                lr = device_bicubic(gt, scale=1.0 / config.DATA.UPSCALE_FACTOR)
            elif fits and engine.lr.shape == lr.shape:
                lr = engine.lr.copy_(lr, non_blocking=True)
            else:
                lr = lr.to(device=config.DEVICE, non_blocking=True)
            loss_values, d_now = engine.step(gt, lr)
            d_loss = d_now if d_now is not None else d_loss
            if batch_num % config.LOG_TRAIN_PERIOD != 0 or rank != 0:
                continue
            batches_done = batch_num + epoch * len(train_loader)
            vals = {k: float(v) for k, v in loss_values.items()}
            writer.add_scalar("Train/D_Loss", float(d_loss), batches_done)
            writer.add_scalar("Train/G_Loss", sum(vals.values()), batches_done)
            for name, v in vals.items():
                writer.add_scalar(f"Train/G_{name}", v, batches_done)
            writer.add_scalar("Train/D(GT)_Probability", torch.sigmoid(torch.mean(engine.pred_gt)).item(), batches_done)
            writer.add_scalar("Train/D(SR)_Probability", torch.sigmoid(torch.mean(engine.pred_sr)).item(), batches_done)
            print(f"[Epoch {epoch+1}/{config.EXP.N_EPOCHS}] [Batch {batch_num}/{len(train_loader)}] "
                  f"[D loss: {float(d_loss)}] [G loss: {sum(vals.values())}] [G losses: {vals}]")
        g_scheduler.step()
        d_scheduler.step()
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
            torch.save(discriminator.state_dict(), results_dir + "/d_last.pth")
            if best_psnr < psnr and best_ssim < ssim:
                torch.save(generator.state_dict(), results_dir + "/g_best.pth")
                torch.save(discriminator.state_dict(), results_dir + "/d_best.pth")
                best_psnr, best_ssim = psnr, ssim
            if 0 < epoch and epoch % config.G_CHECKPOINT_INTERVAL == 0:
                torch.save(generator.state_dict(), results_dir + f"/g_epoch{epoch}.pth")
            if 0 < epoch and epoch % config.D_CHECKPOINT_INTERVAL == 0:
                torch.save(discriminator.state_dict(), results_dir + f"/d_epoch{epoch}.pth")
    engine.close()
    return generator, discriminator


if __name__ == "__main__":
    train(Config())

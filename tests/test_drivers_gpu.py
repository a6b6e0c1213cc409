"""GPU: the warmup()/train() drivers end to end on a synthetic dataset - step loop, LR scheduler under hipGraph,
validation (eval-mode generator on arbitrary image sizes, PSNR/SSIM), checkpoints with the reference's key names."""
import os

import pytest
import torch
from torch.utils.data import Dataset

pytestmark = pytest.mark.gpu


class _Pairs(Dataset):
    """Set5-shaped validation pairs: (gt [3,H,W], lr [3,H/4,W/4]) of different sizes, batch 1."""

    def __init__(self):
        from srganst.bicubic import Bicubic
        g = torch.Generator().manual_seed(9)
        self.items = []
        for h, w in ((48, 64), (72, 40), (96, 96)):
            base = torch.rand(1, 3, h // 8, w // 8, generator=g)
            gt = torch.nn.functional.interpolate(base, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
            gt = torch.round(gt * 255) / 255
            self.items.append((gt[0], Bicubic("cpu")(gt, scale=0.25)[0]))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _cfg(tmp, name):
    from srganst.config import Config
    cfg = Config()
    cfg.EXP.NAME = name
    cfg.EXP.N_EPOCHS = 2
    cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB, cfg.MODEL.D_N_CHANNEL = 16, 2, 8
    cfg.DATA.BATCH_SIZE = 4
    cfg.LOG_TRAIN_PERIOD = 2
    cfg.DATA.TEST_SR_IMAGES_DIR = os.path.join(tmp, "sr")
    return cfg


def test_warmup_and_train_drivers(tmp_path, monkeypatch):
    from srganst.dataset import SyntheticImageDataset
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    from srganst.train import train
    from srganst.utils import load_state_dict
    from srganst.validate import test as run_test
    from srganst.warmup import warmup
    monkeypatch.chdir(tmp_path)
    train_ds = SyntheticImageDataset(24, hr=96, seed=1)
    cfg = _cfg(str(tmp_path), "warm")
    cfg.MODEL.G_LOSS.WARMUP_CRITERIONS = {"Pixel": MSELoss(), "ST": StructureTensorLoss()}
    cfg.MODEL.G_LOSS.WARMUP_WEIGHTS = {"Pixel": 1.0, "ST": 1 / 3}
    G = warmup(cfg, train_dataset=train_ds, test_dataset=_Pairs(), max_steps_per_epoch=5)
    assert os.path.exists("results/warm/g_last.pth")
    sd = torch.load("results/warm/g_last.pth", map_location="cpu", weights_only=True)
    assert "trunk.1.rcb.3.weight" in sd and "upsampling.1.upsample_block.2.weight" in sd and "conv3.bias" in sd
    assert int(sd["trunk.0.rcb.1.num_batches_tracked"]) == 10                  # 2 epochs x 5 steps, one BN bump per step
    G2 = load_state_dict(Generator(cfg), {"_orig_mod." + k: v for k, v in sd.items()})   # reference-style compiled checkpoint
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(G.state_dict().values(), G2.state_dict().values()))

    cfg2 = _cfg(str(tmp_path), "gan")
    cfg2.MODEL.G_CONTINUE_FROM_WARMUP = True
    cfg2.MODEL.G_WARMUP_WEIGHTS = "results/warm/g_last.pth"
    cfg2.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg2.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg2.SOLVER.D_UPDATE_INTERVAL = 2
    cfg2.KERNEL.LR_ON_DEVICE = True          # LR batches synthesised on the GPU from the GT batches (sst_bicubic)
    G3, D3 = train(cfg2, train_dataset=train_ds, test_dataset=_Pairs(), max_steps_per_epoch=5)
    for f in ("g_last.pth", "d_last.pth"):
        assert os.path.exists(os.path.join("results/gan", f))
    dsd = torch.load("results/gan/d_last.pth", map_location="cpu", weights_only=True)
    assert "features.20.weight" in dsd and "classifier.2.bias" in dsd
    assert set(dsd.keys()) == set(Discriminator(cfg2).state_dict().keys())
    # validate.test(): loads g_best/g_last, runs batch-1 eval, writes _metrics.txt
    psnr, ssim = run_test(cfg2, save_images=False, g_path="results/gan/g_last.pth", dataset=_Pairs())
    assert 5.0 < psnr < 60.0 and -1.0 <= ssim <= 1.0
    assert os.path.exists(os.path.join(cfg2.DATA.TEST_SR_IMAGES_DIR, "gan", "_metrics.txt"))


def test_bicubic_on_device_matches_reference_golden_and_host_path():
    """sst_bicubic (LR synthesis on the GPU) against the reference's own outputs (tests/golden/bicubic.npz) and
    against the host path the data loader uses - equal on the 1/255 grid."""
    import numpy as np
    from srganst.bicubic import Bicubic
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bicubic.npz"))
    hr = torch.from_numpy(g["hr_u8"]).float() / 255
    out = Bicubic("cuda")(hr.cuda(), scale=0.25)
    assert out.shape == (1, 3, 24, 24) and torch.equal(out.cpu(), torch.from_numpy(g["lr"]))
    step = torch.zeros(1, 3, 96, 96)
    step[..., 48:] = 1.0
    s = Bicubic("cuda")(step.cuda(), scale=0.25).cpu()
    assert torch.equal(s, torch.from_numpy(g["step_lr"])) and float(s.min()) < 0 and float(s.max()) > 1
    gen = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (5, 3, 192, 96), generator=gen).float() / 255          # non-square, batch > 1
    a, b = Bicubic("cpu")(x, scale=0.25), Bicubic("cuda")(x.cuda(), scale=0.25).cpu()
    # identical except where 255*x lands within float rounding of a half-way point (summation order differs)
    assert float((a - b).abs().max()) <= 1.0 / 255 + 1e-7 and float((a != b).float().mean()) < 1e-3

"""ContentLossDiscriminator (reference loss.py:231-289): oracle vs the fixture generated with the reference's own
Discriminator (tests/golden/make_golden_dfeat.py), HIP path vs both."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

G = os.path.join(os.path.dirname(__file__), "golden", "disc_content.npz")
LAYERS = {"features.4": 0.25, "features.10": 0.5}


def _sd(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


@pytest.mark.parametrize("crit", ["mse", "l1"])
def test_oracle_matches_reference_golden(crit):
    from oracle import model as om
    g = np.load(G)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    gt = torch.from_numpy(g["gt"])
    loss = om.disc_content_loss(_sd(g), x, gt, LAYERS, crit)
    (gx,) = torch.autograd.grad(loss, x)
    assert abs(float(loss) - float(g[f"{crit}/loss"])) < 1e-7 * abs(float(g[f"{crit}/loss"]))
    assert rel_err(gx, torch.from_numpy(g[f"{crit}/grad"])) < 1e-6
    mean = torch.tensor(om.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(om.IMAGENET_STD).view(1, 3, 1, 1)
    f = om.discriminator_features(_sd(g), (x.detach() - mean) / std, {4, 10})
    assert rel_err(f[4], torch.from_numpy(g["feat4"])) < 1e-6 and rel_err(f[10], torch.from_numpy(g["feat10"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("crit", ["mse", "l1"])
def test_hip_disc_content_matches_reference_golden(crit):
    from srganst.config import Config
    from srganst.loss import ContentLossDiscriminator
    g = np.load(G)
    cfg = Config()
    cfg.MODEL.D_N_CHANNEL = 16
    cfg.DEVICE = "cuda"
    mod = ContentLossDiscriminator(cfg, criterion=crit)
    missing = mod.D.load_state_dict({k: v for k, v in _sd(g).items()}, strict=False)
    assert not missing.unexpected_keys
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(g["gt"]).cuda()
    loss = mod(x, gt)
    (gx,) = torch.autograd.grad(loss * 3.0, x)
    ref = float(g[f"{crit}/loss"])
    assert abs(loss.item() - ref) < 1e-4 * abs(ref)
    tol = 1e-3 if crit == "mse" else 2e-2          # L1: a sign flip of a near-zero feature difference moves one element by 2/N
    assert rel_err(gx.cpu() / 3.0, torch.from_numpy(g[f"{crit}/grad"])) < tol


@pytest.mark.gpu
def test_hip_disc_content_full_width_vs_oracle():
    from oracle import model as om
    from srganst.config import Config
    from srganst.loss import ContentLossDiscriminator
    cfg = Config()
    cfg.DEVICE = "cuda"
    torch.manual_seed(21)
    mod = ContentLossDiscriminator(cfg)                       # 64 channels, reference default taps
    sd = {k: v.detach().cpu() for k, v in mod.D.state_dict().items()}
    gen = torch.Generator().manual_seed(22)
    gt = torch.rand(2, 3, 96, 96, generator=gen)
    x = (gt + 0.1 * torch.randn(gt.shape, generator=gen)).clamp(0, 1)
    x64 = x.double().requires_grad_(True)
    l64 = om.disc_content_loss({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, x64, gt.double(),
                               cfg.MODEL.G_LOSS.DISC_FEATURES_LOSS_LAYERS)
    l64.backward()
    xg = x.cuda().requires_grad_(True)
    loss = mod(xg, gt.cuda())
    loss.backward()
    assert abs(loss.item() - l64.item()) < 1e-4 * abs(l64.item())
    assert rel_err(xg.grad.cpu(), x64.grad) < 1e-3

#!/usr/bin/env python3
"""Golden fixture for ContentLossDiscriminator (reference loss.py:231-289).  The reference class needs torchvision's
create_feature_extractor / transforms.Normalize, which are not installed; the fixture is therefore produced with the reference's
own model.Discriminator (seeded, eval mode, `features[:idx+1]` is what the extractor returns for node "features.<idx>") and the
documented Normalize formula (x - mean) / std.  Build container only.  Re-run: python tests/golden/make_golden_dfeat.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, lowfreq, save  # noqa: E402


def main():
    rconfig, rmodel, _, _, _ = import_reference()
    cfg = rconfig.Config()
    cfg.MODEL.D_N_CHANNEL = 16                      # reduced width: small fixture (the full width is checked against the oracle)
    cfg.DEVICE = "cpu"
    torch.manual_seed(7)
    D = rmodel.Discriminator(cfg).eval()
    # make the eval-mode BatchNorms non-trivial
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for m in D.features:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    gen = torch.Generator().manual_seed(9)
    gt = lowfreq(gen, 2, 48)
    x = (gt + 0.08 * torch.randn(gt.shape, generator=gen)).clamp(0, 1).requires_grad_(True)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    layers = {"features.4": 0.25, "features.10": 0.5}
    arrs = {"x": x.detach().numpy(), "gt": gt.numpy()}
    for crit_name, crit in (("mse", torch.nn.MSELoss()), ("l1", torch.nn.L1Loss())):
        loss = torch.tensor(0.0)
        for name, w in layers.items():
            idx = int(name.split(".")[1])
            fx = D.features[:idx + 1]((x - mean) / std)
            fg = D.features[:idx + 1]((gt - mean) / std)
            loss = loss + w * crit(fx, fg)
            if crit_name == "mse":
                arrs[f"feat{idx}"] = fx.detach().numpy()
        (gx,) = torch.autograd.grad(loss, x)
        arrs[f"{crit_name}/loss"] = loss.detach().numpy()
        arrs[f"{crit_name}/grad"] = gx.numpy()
    for k, v in D.state_dict().items():
        if k.startswith("features.") and int(k.split(".")[1]) <= 9:      # the layers the two taps need
            arrs["sd/" + k] = v.numpy()
    save("disc_content", **arrs)


if __name__ == "__main__":
    main()

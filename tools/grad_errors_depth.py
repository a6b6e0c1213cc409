#!/usr/bin/env python3
"""Dev tool: generator-only gradient errors (pixel MSE, B = 16) vs the fp64 oracle for several trunk depths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from conftest import oracle_grads, rel_err
from oracle import model as om
from srganst.config import Config
from srganst.loss import MSELoss
from srganst.model import Generator

for depth in (1, 2, 4, 16):
    cfg = Config()
    cfg.MODEL.G_N_RCB = depth
    torch.manual_seed(41)
    G = Generator(cfg)
    sd0 = {k: v.clone() for k, v in G.state_dict().items()}
    gen = torch.Generator().manual_seed(42)
    lr, gt = torch.rand(16, 3, 24, 24, generator=gen), torch.rand(16, 3, 96, 96, generator=gen)
    fl = lambda sdx, lr_, gt_: F.mse_loss(om.generator_forward(sdx, lr_, True, {}), gt_)
    ins = ((lr, False), (gt, False))
    _, g32, _, _ = oracle_grads(fl, sd0, torch.float32, ins)
    _, g64, _, _ = oracle_grads(fl, sd0, torch.float64, ins)
    G.cuda().train()
    MSELoss()(G(lr.cuda()), gt.cuda()).backward()
    rows = [(rel_err(p.grad.cpu(), g64[n]), rel_err(g32[n], g64[n]), n) for n, p in G.named_parameters()]
    print(f"depth {depth}:")
    for r in sorted(rows, key=lambda r: -r[0] / max(1e-3, 3 * r[1]))[:6]:
        print(f"  hip {r[0]:.2e}  ref {r[1]:.2e}  {r[2]}")

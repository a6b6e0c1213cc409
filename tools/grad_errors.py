#!/usr/bin/env python3
"""Dev tool: per-parameter gradient error of the full-size generator / discriminator on the HIP path against the oracle in fp64,
next to the error of the oracle's own fp32 run (the fp64-truth rule of tests/conftest.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from conftest import oracle_grads, rel_err
from oracle import model as om
from srganst.config import Config
from srganst.loss import BCEWithLogitsLoss, MSELoss
from srganst.model import Discriminator, Generator

cfg = Config()
torch.manual_seed(0)
G = Generator(cfg)
sd0 = {k: v.clone() for k, v in G.state_dict().items()}
gen = torch.Generator().manual_seed(1)
lr, gt = torch.rand(16, 3, 24, 24, generator=gen), torch.rand(16, 3, 96, 96, generator=gen)
fl = lambda sdx, lr_, gt_: F.mse_loss(om.generator_forward(sdx, lr_, True, {}), gt_)
ins = ((lr, False), (gt, False))
_, g32, _, _ = oracle_grads(fl, sd0, torch.float32, ins)
_, g64, _, _ = oracle_grads(fl, sd0, torch.float64, ins)
G.cuda().train()
MSELoss()(G(lr.cuda()), gt.cuda()).backward()
rows = [(rel_err(p.grad.cpu(), g64[n]), rel_err(g32[n], g64[n]), n) for n, p in G.named_parameters()]
print("generator B=16: worst HIP errors vs fp64 (hip, oracle fp32, name)")
for r in sorted(rows, reverse=True)[:12]:
    print(f"  {r[0]:.2e} {r[1]:.2e} {r[2]}")
print("  scalar (PReLU) params:")
for r in sorted([r for r in rows if "rcb.2" in r[2] or r[2].endswith(".1.weight") and "conv1" in r[2] or "upsample_block.2" in r[2]], reverse=True)[:8]:
    print(f"  {r[0]:.2e} {r[1]:.2e} {r[2]}")
torch.manual_seed(0)
D = Discriminator(cfg)
sd0 = {k: v.clone() for k, v in D.state_dict().items()}
x = torch.rand(16, 3, 96, 96, generator=gen)


def fd(sdx, x_):
    lg = om.discriminator_forward(sdx, x_, True, {})
    return F.binary_cross_entropy_with_logits(lg, torch.full_like(lg, 0.9))


_, d32, _, _ = oracle_grads(fd, sd0, torch.float32, ((x, True),))
_, d64, _, _ = oracle_grads(fd, sd0, torch.float64, ((x, True),))
D.cuda().train()
BCEWithLogitsLoss()(D(x.cuda()), torch.full([16, 1], 0.9).cuda()).backward()
rows = [(rel_err(p.grad.cpu(), d64[n]), rel_err(d32[n], d64[n]), n) for n, p in D.named_parameters()]
print("discriminator B=16: worst HIP errors vs fp64 (hip, oracle fp32, name)")
for r in sorted(rows, reverse=True)[:10]:
    print(f"  {r[0]:.2e} {r[1]:.2e} {r[2]}")

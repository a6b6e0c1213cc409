#!/usr/bin/env python3
"""Dev tool: run the small SRGAN train loop eager/eager/graph/graph and report max |diff| per parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst.config import Config
from srganst.engine import TrainEngine
from srganst.loss import MSELoss, StructureTensorLoss
from srganst.model import Discriminator, Generator

def run(use_graph, steps=5):
    cfg = Config(); cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB, cfg.MODEL.D_N_CHANNEL = 16, 2, 8
    torch.manual_seed(1)
    D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0); cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=use_graph)
    gen = torch.Generator().manual_seed(2)
    for _ in range(steps):
        eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
    torch.cuda.synchronize()
    out = {"g." + k: v.clone() for k, v in G.state_dict().items()}
    out.update({"d." + k: v.clone() for k, v in D.state_dict().items()})
    out.update({"dgrad." + n: p.grad.clone() for n, p in D.named_parameters()})
    return out

def cmp(a, b, tag):
    for pre in ("g.", "d.", "dgrad."):
        worst = sorted(((float((a[k].float() - b[k].float()).abs().max() / (b[k].float().abs().max() + 1e-30)), k) for k in a if k.startswith(pre)), reverse=True)[:3]
        print(tag, pre, [(f"{v:.1e}", k) for v, k in worst])

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for st in range(2, steps + 1):
    cmp(run(False, st), run(True, st), f"steps={st} eager vs graph:")

#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst.config import Config
from srganst.engine import _GraphedStep, make_adam
from srganst.loss import BCEWithLogitsLoss
from srganst.model import Discriminator

def run(use_graph, steps, capt):
    cfg = Config(); cfg.MODEL.D_N_CHANNEL = 8
    torch.manual_seed(1)
    D = Discriminator(cfg).cuda().train()
    opt = make_adam(D.parameters(), 1e-4, (0.9, 0.999), 1e-4, 0, capturable=capt)
    adv = BCEWithLogitsLoss()
    gen = torch.Generator().manual_seed(2)
    a = torch.rand(4, 3, 96, 96, generator=gen).cuda(); b = torch.rand(4, 3, 96, 96, generator=gen).cuda()
    sa, sb = a.clone(), b.clone()
    def fn():
        opt.zero_grad(set_to_none=True)
        l = adv(D(sa), 0.9) + adv(D(sb.detach().clone()), 0.0)
        l.backward()
        opt.step()
        return l.detach()
    gs = _GraphedStep(fn, enabled=use_graph)
    for i in range(steps):
        sa.copy_(torch.rand(4, 3, 96, 96, generator=gen).cuda()); sb.copy_(torch.rand(4, 3, 96, 96, generator=gen).cuda())
        gs()
    torch.cuda.synchronize()
    out = {"d." + k: v.clone() for k, v in D.state_dict().items()}
    out.update({"dgrad." + n: p.grad.clone() for n, p in D.named_parameters()})
    return out

def cmp(a, b, tag):
    for pre in ("d.", "dgrad."):
        worst = sorted(((float((a[k].float() - b[k].float()).abs().max() / (b[k].float().abs().max() + 1e-30)), k) for k in a if k.startswith(pre)), reverse=True)[:3]
        print(tag, pre, [(f"{v:.1e}", k) for v, k in worst])

for st in (2, 3, 4):
    cmp(run(False, st, True), run(True, st, True), f"D-only steps={st} eager(capturable) vs graph:")

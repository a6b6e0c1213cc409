"""dev: per-step / per-parameter difference of the batched discriminator step against the pass-by-pass one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")]
import torch
from conftest import rel_err
from srganst.config import Config
from srganst.engine import TrainEngine
from srganst.loss import MSELoss, StructureTensorLoss
from srganst.model import Discriminator, Generator


def run(batched, steps):
    cfg = Config()
    cfg.MODEL.G_N_RCB = 2
    cfg.KERNEL.REUSE_D_SR, cfg.KERNEL.BATCH_D_STEP = False, batched
    cfg.KERNEL.DEFER_D_WGRAD = int(os.environ.get("DEFER", "8"))
    torch.manual_seed(1)
    D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=False, adam_capturable=True)
    gen = torch.Generator().manual_seed(2)
    out = []
    for _ in range(steps):
        eng.step(torch.rand(8, 3, 96, 96, generator=gen).cuda(), torch.rand(8, 3, 24, 24, generator=gen).cuda())
        torch.cuda.synchronize()
        out.append(({n: p.grad.clone() for n, p in D.named_parameters()}, {n: p.detach().clone() for n, p in D.named_parameters()},
                    {n: p.grad.clone() for n, p in G.named_parameters()}, eng.sr.clone()))
    return out


a, b = run(False, 4), run(True, 4)
for i, ((ga, pa, gga, sra), (gb, pb, ggb, srb)) in enumerate(zip(a, b)):
    worst = sorted(((rel_err(gb[n], ga[n]), n) for n in ga), reverse=True)[:4]
    print(f"step {i}: sr {rel_err(srb, sra):.2e}  G grads {max(rel_err(ggb[n], gga[n]) for n in gga):.2e}  D params "
          f"{max(rel_err(pb[n], pa[n]) for n in pa):.2e}  D grads worst {worst}")

#!/usr/bin/env python3
"""Dev tool: per-parameter gradient error of ONE full-size SRGAN iteration (B = 16, adv + pixel + ST; generator and discriminator
step) on the HIP path against oracle/steps.py in fp64, next to the error of the oracle's own fp32 run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import rel_err
from oracle import steps as osteps
from srganst.config import Config
from srganst.engine import TrainEngine
from srganst.loss import MSELoss, StructureTensorLoss
from srganst.model import Discriminator, Generator

which = sys.argv[1] if len(sys.argv) > 1 else "all"
crits = {"all": (("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0)), "nost": (("Adversarial", 0.001), ("Pixel", 1.0)),
         "pix": (("Pixel", 1.0),), "st": (("ST", 1.0 / 3.0),)}[which]
cfg = Config()
torch.manual_seed(41)
D, G = Discriminator(cfg), Generator(cfg)
g0 = {k: v.clone() for k, v in G.state_dict().items()}
d0 = {k: v.clone() for k, v in D.state_dict().items()}
gen = torch.Generator().manual_seed(int(os.environ.get("SST_GE_SEED", "42")))
gt, lr = torch.rand(16, 3, 96, 96, generator=gen), torch.rand(16, 3, 24, 24, generator=gen)


srs = []


def oracle_iter(dtype, device="cpu"):
    cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v).to(device) for k, v in sd.items()}
    tr = osteps.OracleTrainer(cast(g0), cast(d0), criterions=crits, d_update_interval=1)
    sr = tr.train_step(gt.to(dtype).to(device), lr.to(dtype).to(device))[0]
    cpu = lambda d: {k: v.cpu() for k, v in d.items()}
    srs.append(sr.cpu())
    return cpu(tr.g_grads()), cpu(tr.d_grads())


gg32, dg32 = oracle_iter(torch.float32)
gg64, dg64 = oracle_iter(torch.float64)
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
ggm, dgm = oracle_iter(torch.float32, "cuda")        # third opinion: the same plain-torch graph on the GPU (MIOpen / rocBLAS, fp32)
D.cuda().train(); G.cuda().train()
for name, w in crits:
    if name == "Pixel":
        cfg.add_g_criterion("Pixel", MSELoss(), w)
    if name == "ST":
        cfg.add_g_criterion("ST", StructureTensorLoss(), w)
if "Adversarial" not in dict(crits):
    cfg.remove_g_criterion("Adversarial")
cfg.SOLVER.D_UPDATE_INTERVAL = 1
eng = TrainEngine(cfg, G, D, use_graph=False)
eng.step(gt.cuda(), lr.cuda())
flips = lambda a, b: int(((a == 0) != (b == 0)).sum() + ((a == 1) != (b == 1)).sum())
print(f"seed {os.environ.get('SST_GE_SEED', '42')}: clamp-mask flips against the fp64 run: hip {flips(eng.sr.cpu(), srs[1])}, oracle fp32 {flips(srs[0], srs[1])}, "
      f"torch-gpu {flips(srs[2], srs[1])}")
rows = [(rel_err(p.grad.cpu(), gg64[n]), rel_err(gg32[n], gg64[n]), "G." + n, rel_err(ggm[n], gg64[n])) for n, p in G.named_parameters()]
rows += [(rel_err(p.grad.cpu(), dg64[n]), rel_err(dg32[n], dg64[n]), "D." + n, rel_err(dgm[n], dg64[n])) for n, p in D.named_parameters()]
badm = [r for r in rows if r[3] > max(1e-3, 3 * r[1])]
import statistics
print(f"torch-GPU fp32 (MIOpen): {len(badm)} of {len(rows)} beyond the same bound; median err ratio hip/cpu {statistics.median(r[0] / max(r[1], 1e-12) for r in rows):.2f}, "
      f"torch-gpu/cpu {statistics.median(r[3] / max(r[1], 1e-12) for r in rows):.2f}, hip/torch-gpu {statistics.median(r[0] / max(r[3], 1e-12) for r in rows):.2f}")
bad = [r for r in rows if r[0] > max(1e-3, 3 * r[1])]
print(f"criterions {which}: {len(bad)} of {len(rows)} parameters beyond max(1e-3, 3 x reference error)")
for r in sorted(rows, key=lambda r: -r[0] / max(1e-3, 3 * r[1]))[:25]:
    print(f"  hip {r[0]:.2e}  ref {r[1]:.2e}  torch-gpu {r[3]:.2e}  ratio-to-bound {r[0] / max(1e-3, 3 * r[1]):.2f}  {r[2]}")

if os.environ.get("SST_GE_ALL"):
    print("-- all generator parameters in network order")
    for r in rows:
        if r[2].startswith("G."):
            print(f"  hip {r[0]:.2e}  ref {r[1]:.2e}  torch-gpu {r[3]:.2e}  {r[2]}")

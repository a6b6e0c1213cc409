#!/usr/bin/env python3
"""Dev tool: forward accuracy of the full-size generator (train mode, B = 16) on the HIP path against the oracle in fp64, next to the
oracle's own fp32 run: SR and the batch statistics of every BatchNorm (read back from the running buffers after one forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "srgan-st_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import rel_err
from oracle import model as om
from srganst.config import Config
from srganst.model import Generator

cfg = Config()
torch.manual_seed(41)
G = Generator(cfg)
g0 = {k: v.clone() for k, v in G.state_dict().items()}
gen = torch.Generator().manual_seed(42)
lr = torch.rand(16, 3, 24, 24, generator=gen)
res = {}
for dt in (torch.float32, torch.float64):
    sd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in g0.items()}
    nb = {}
    with torch.no_grad():
        sr = om.generator_forward(sd, lr.to(dt), True, nb)
    res[dt] = (sr, nb)
G.cuda().train()
with torch.no_grad():
    sr = G(lr.cuda())
sd1 = G.state_dict()
sr32, nb32 = res[torch.float32]
sr64, nb64 = res[torch.float64]
print(f"SR: hip {rel_err(sr.cpu(), sr64):.2e}  oracle fp32 {rel_err(sr32, sr64):.2e}")
rows = []
for k in nb64:
    if "num_batches" in k:
        continue
    rows.append((rel_err(sd1[k].cpu(), nb64[k]), rel_err(nb32[k], nb64[k]), k))
for r in rows[::6] + sorted(rows, reverse=True)[:6]:
    print(f"  hip {r[0]:.2e}  ref {r[1]:.2e}  {r[2]}")

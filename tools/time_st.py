#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the ST-loss forward / backward kernels at the bench size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst.loss import StructureTensorLoss
from ablate_wgrad import timeit  # noqa

B, H = 16, 96
sr = torch.rand(B, 3, H, H, device="cuda", requires_grad=True)
gt = torch.rand(B, 3, H, H, device="cuda")
crit = StructureTensorLoss()
def fwd():
    with torch.no_grad():
        return crit(sr, gt)
def fb():
    sr.grad = None
    crit(sr, gt).backward()
print("fwd", timeit(fwd), "us   fwd+bwd (incl. autograd glue)", timeit(fb), "us")

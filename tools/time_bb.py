#!/usr/bin/env python3
"""Dev tool: GPU-bound time of BestBuddyLoss forward+backward at the bench size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst.loss import BestBuddyLoss
from ablate_wgrad import timeit  # noqa
B, H = 16, 96
sr = torch.rand(B, 3, H, H, device="cuda", requires_grad=True)
gt = torch.rand(B, 3, H, H, device="cuda")
crit = BestBuddyLoss()
def fb():
    sr.grad = None
    crit(sr, gt).backward()
print("BestBuddyLoss fwd+bwd B=16 96px: %.1f us" % timeit(fb))

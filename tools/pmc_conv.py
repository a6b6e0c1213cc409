#!/usr/bin/env python3
"""Dev tool: launch the trunk / up-conv forward kernels a few times (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops
for (B, H, W, Cin, Cout) in [(16, 24, 24, 64, 64), (16, 48, 48, 64, 256)]:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    for _ in range(20):
        ops.conv_fwd(x, wp, Cout, 3, 1)
    dy = torch.randn(B, H, W, Cout, device="cuda")
    dw = torch.empty_like(w)
    for _ in range(10):
        ops.conv_wgrad(x, dy, dw, 3, 1)
torch.cuda.synchronize()

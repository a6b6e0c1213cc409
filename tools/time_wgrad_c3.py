#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the first discriminator layer's weight gradient (3 -> 64, B = 16), MFMA vs VALU kernel, incl. the slab reduce."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit
for (B, H) in [(16, 96), (8, 192)]:
    x = torch.randn(B, H, H, 3, device="cuda")
    dy = torch.randn(B, H, H, 64, device="cuda")
    dw = torch.empty(64, 3, 3, 3, device="cuda")
    row = f"B{B} {H}px 3->64: dY {dy.numel()*4/1e6:.1f} MB"
    for mode in ("mfma", "valu"):
        if mode == "valu":
            os.environ["SST_WGRAD_NO_K3C3_MFMA"] = "1"
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 1))
        row += f" | {mode} {t:6.1f} us ({dy.numel()*4/t/1e6:5.2f} TB/s)"
    os.environ.pop("SST_WGRAD_NO_K3C3_MFMA")
    print(row, flush=True)

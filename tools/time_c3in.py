#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the first discriminator layer's forward (3 -> 64 + bias), MFMA vs VALU kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit
for (B, H) in [(16, 96), (8, 192)]:
    x = torch.randn(B, H, H, 3, device="cuda")
    w = torch.randn(64, 3, 3, 3, device="cuda")
    b = torch.randn(64, device="cuda")
    wp = ops.pack_conv(w)
    mb = B * H * H * 64 * 4 / 1e6
    row = f"B{B} {H}px 3->64: output {mb:.1f} MB"
    for mode in ("mfma", "valu"):
        if mode == "valu":
            os.environ["SST_NO_C3IN_MFMA"] = "1"
        t = timeit(lambda: ops.conv_fwd(x, wp, 64, 3, 1, bias=b))
        row += f" | {mode} {t:6.1f} us ({mb / t:5.2f} TB/s)"
    os.environ.pop("SST_NO_C3IN_MFMA")
    print(row, flush=True)

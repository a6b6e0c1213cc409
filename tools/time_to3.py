#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the data-gradient of the first discriminator layer (64 -> 3 channels, B = 16): folded-kx kernel vs general kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit
for (B, H) in [(16, 96), (8, 192)]:
    dy = torch.randn(B, H, H, 64, device="cuda")
    w = torch.randn(64, 3, 3, 3, device="cuda")
    wd = ops.pack_conv(w, 1)
    row = f"B{B} {H}px 64->3 dgrad: dY {dy.numel()*4/1e6:.1f} MB"
    for mode in ("folded", "general"):
        if mode == "general":
            os.environ["SST_NO_TO3"] = "1"
        t = timeit(lambda: ops.conv_fwd(dy, wd, 3, 3, 1))
        row += f" | {mode} {t:6.1f} us"
    os.environ.pop("SST_NO_TO3")
    print(row, flush=True)

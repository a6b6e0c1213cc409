#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the discriminator's stride-2 weight gradients (B = 16): all-taps kernel vs per-tap kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, c) in [(96, 64), (48, 128), (24, 256), (12, 512)]:
    x = torch.randn(B, H, H, c, device="cuda")
    dy = torch.randn(B, H // 2, H // 2, c, device="cuda")
    dw = torch.empty(c, c, 3, 3, device="cuda")
    sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    fl = 2.0 * B * (H // 2) ** 2 * c * c * 9
    row = f"B{B} {H:3d}px {c:3d}->{c:3d} s2 ({fl/1e9:5.2f} GF, ideal {fl/157.3e6:5.1f} us):"
    for mode in ("1", "0"):
        os.environ["SST_WGRAD_S2"] = mode
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 2, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1))
        row += f"  {'all-taps' if mode == '1' else 'per-tap '} {t:6.1f} us {fl/t/1e6:5.1f} TF"
    os.environ.pop("SST_WGRAD_S2")
    print(row, flush=True)

#!/usr/bin/env python3
"""Dev tool: stride-2 data-gradient of the discriminator's layers, pipelined kernel vs conv_s2dgrad4_kernel (B = 16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit
B = 16
for (H, cin, cout) in [(96, 64, 64), (48, 128, 128), (24, 256, 256), (12, 512, 512)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv_s2_dgrad(w)
    dy = torch.randn(B, H // 2, H // 2, cout, device="cuda")
    fl = 2.0 * B * (H // 2) ** 2 * cin * cout * 9
    row = f"B{B} dX {H:3d}px {cout:3d}->{cin:3d} ({fl/1e9:5.2f} GF, ideal {fl/157.3e6:5.1f} us):"
    for mode in ("1", "0"):
        os.environ["SST_CONV_PIPE"] = mode
        t = timeit(lambda: ops.conv_s2_dgrad(dy, wp, H, H, cin))
        row += f"  {'pipe' if mode == '1' else 's2dgrad4'} {t:6.1f} us {fl/t/1e6:5.1f} TF"
    print(row, flush=True)

#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the discriminator's conv shapes (fwd, dgrad, wgrad), B = 16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops
from ablate_wgrad import timeit  # noqa

B = 16
for (H, cin, cout, s) in [(96, 3, 64, 1), (96, 64, 64, 2), (48, 64, 128, 1), (48, 128, 128, 2), (24, 128, 256, 1), (24, 256, 256, 2),
                          (12, 256, 512, 1), (12, 512, 512, 2)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    ho = (H + 2 - 3) // s + 1
    dy = torch.randn(B, ho, ho, cout, device="cuda")
    dw = torch.empty_like(w)
    fl = 2.0 * B * ho * ho * cin * cout * 9
    wp = ops.pack_conv(w)
    tf = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s))
    if s == 1:
        wd = ops.pack_conv(w, 1)
        td = timeit(lambda: ops.conv_fwd(dy, wd, cin, 3, 1)) if cin % 4 == 0 else float("nan")
    else:
        wd = ops.pack_conv_s2_dgrad(w)
        td = timeit(lambda: ops.conv_s2_dgrad(dy, wd, H, H, cin))
    tw = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, s))
    print(f"{H:3d}px {cin:3d}->{cout:3d} s{s}: {fl/1e9:5.2f} GF | fwd {tf:6.1f} us {fl/tf/1e6:5.1f} TF | dgrad {td:6.1f} us {fl/td/1e6:5.1f} TF | wgrad {tw:6.1f} us {fl/tw/1e6:5.1f} TF")

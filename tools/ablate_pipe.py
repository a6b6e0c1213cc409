#!/usr/bin/env python3
"""Dev tool: phase ablation of the pipelined conv kernel (SST_PIPE_DBG bits: 1 no LDS staging writes, 2 no epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit

B = 16
for (H, cin, cout, s) in [(48, 64, 128, 1), (24, 128, 256, 1), (12, 256, 512, 1), (48, 128, 128, 2), (12, 512, 512, 2)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho = (H - 1) // s + 1
    fl = 2.0 * B * ho * ho * cin * cout * 9
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    row = f"B{B} {H:3d}px {cin:3d}->{cout:3d} s{s} (ideal {fl/157.3e6:5.1f} us):"
    for name, dbg in [("full", 0), ("no-stage-store", 1), ("no-epi", 2), ("kloop only", 3)]:
        os.environ["SST_PIPE_DBG"] = str(dbg)
        t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
        row += f" | {name} {t:6.1f}"
    os.environ["SST_PIPE_DBG"] = "0"
    print(row, flush=True)

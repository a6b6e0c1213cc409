#!/usr/bin/env python3
"""Dev tool: what would the N-split kernel do on the layers that have too few units at the step's batch sizes?  The same layers at
batch sizes that give them >= 1,152 units, N-split (SST_CONV_NS=1) against K-split (0): forward with input affine + LeakyReLU +
statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from time_pipe_lib import timeit  # noqa: E402

for (B, H, cin, cout, s) in [(64, 48, 128, 128, 2), (128, 24, 256, 256, 2), (256, 12, 512, 512, 2), (64, 12, 256, 512, 1), (32, 24, 128, 256, 1),
                             (64, 24, 256, 128, 1), (64, 12, 512, 256, 1)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho = (H - 1) // s + 1
    fl = 2.0 * B * ho * ho * cin * cout * 9
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    row = f"B{B:3d} {H:3d}px {cin:3d}->{cout:3d} s{s} ({fl/1e9:6.2f} GF):"
    for mode in (True, False):
        ops.CONV_NS = mode
        t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
        row += f"  {'n-split' if mode else 'k-split'} {t:7.1f} us {fl/t/1e6:6.1f} TF"
    print(row, flush=True)

#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the discriminator's conv shapes through the pipelined kernel vs the general one (B = 16):
forward (with input affine + LeakyReLU + statistics) and stride-1 data-gradient (with backward partials)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops


from time_pipe_lib import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, cin, cout, s) in [(96, 64, 64, 2), (48, 64, 128, 1), (48, 128, 128, 2), (24, 128, 256, 1), (24, 256, 256, 2), (12, 256, 512, 1),
                          (12, 512, 512, 2)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho = (H - 1) // s + 1
    fl = 2.0 * B * ho * ho * cin * cout * 9
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    row = f"B{B} {H:3d}px {cin:3d}->{cout:3d} s{s} ({fl/1e9:5.2f} GF, ideal {fl/157.3e6:5.1f} us):"
    for mode in ("1", "0"):
        os.environ["SST_CONV_PIPE"] = mode
        t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
        row += f"  fwd {'pipe' if mode == '1' else 'gen '} {t:6.1f} us {fl/t/1e6:5.1f} TF"
    if s == 1:
        dy = torch.randn(B, H, H, cout, device="cuda")
        wd = ops.pack_conv(w, 1)
        for mode in ("1", "0"):
            os.environ["SST_CONV_PIPE"] = mode
            t = timeit(lambda: ops.conv_dgrad_bwdstats(dy, wd, cin, 3, x, epi_scale=sc, epi_shift=sh, epi_slope_const=0.2, epi_act=1))
            row += f"  | dgrad {'pipe' if mode == '1' else 'gen '} {t:6.1f} us {fl/t/1e6:5.1f} TF"
    print(row, flush=True)

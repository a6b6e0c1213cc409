#!/usr/bin/env python3
"""Dev tool: phase ablation (staging / K loop / epilogue) of the general conv kernel on the discriminator's shapes.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops
from ablate_big import timeit  # noqa

os.environ["SST_CONV_BIG"] = "0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, cin, cout, s) in [(48, 64, 128, 1), (48, 128, 128, 2), (24, 128, 256, 1), (24, 256, 256, 2), (12, 256, 512, 1), (12, 512, 512, 2),
                          (96, 64, 64, 2)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho = (H + 2 - 3) // s + 1
    fl = 2.0 * B * ho * ho * cin * cout * 9
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    row = []
    for name, dbg in [("full", 0), ("no-stage", 1), ("no-kloop", 2), ("no-epi", 4), ("stage only", 6), ("kloop only", 5), ("epi only", 3), ("empty", 7)]:
        t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, out_mode=dbg << 8))
        row.append(f"{name} {t:6.1f}")
    t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
    print(f"B{B} {H:3d}px {cin:3d}->{cout:3d} s{s} ({fl/1e9:5.2f} GF, ideal {fl/157.3e6:5.1f} us): " + " | ".join(row) + f" | full+bn+stats {t:6.1f} us ({fl/t/1e6:5.1f} TF)")

#!/usr/bin/env python3
"""Dev tool: up-sampler-shaped convs (fwd with PixelShuffle store, dgrad with 256 inputs) band vs general kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops
from ablate_wgrad import timeit  # noqa

for (B, H, W, Cin, Cout, mode) in [(16, 24, 24, 64, 256, ops.OUT_SHUFFLE), (16, 48, 48, 64, 256, ops.OUT_SHUFFLE),
                                   (16, 24, 24, 256, 64, ops.OUT_NHWC), (16, 48, 48, 256, 64, ops.OUT_NHWC), (16, 24, 24, 64, 64, ops.OUT_NHWC)]:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    flops = 2.0 * B * H * W * Cin * Cout * 9
    line = f"B{B} {H}x{W} {Cin}->{Cout} mode{mode}: {flops/1e9:5.2f} GFLOP "
    for band, ng in (("0", "0"), ("3", "1"), ("3", "2"), ("3", "4")):
        os.environ["SST_CONV_BAND"] = band
        os.environ["SST_CONV_BAND_NG"] = ng
        t = timeit(lambda: ops.conv_fwd(x, wp, Cout, 3, 1, out_mode=mode))
        line += f"| band={band} ng={ng}: {t:7.1f} us {flops/t/1e6:6.1f} TF/s "
    print(line)

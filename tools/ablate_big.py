#!/usr/bin/env python3
"""Dev tool: 32x32-tile vs 64x64-tile conv kernel on the large shapes (GPU-bound timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops

def timeit(fn, n=30, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

shapes = [(16, 48, 48, 64, 256, 1), (16, 96, 96, 64, 64, 1), (16, 96, 96, 64, 64, 2), (16, 48, 48, 64, 128, 1), (16, 24, 24, 128, 256, 1),
          (16, 12, 12, 256, 512, 1), (16, 12, 12, 512, 512, 2), (32, 96, 96, 64, 64, 1), (32, 24, 24, 256, 256, 1), (16, 24, 24, 64, 64, 1),
          (16, 24, 24, 64, 256, 1)]
for (B, H, W, Cin, Cout, s) in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho, wo = ops.conv_out_hw(H, W, 3, s)
    flops = 2.0 * B * ho * wo * Cin * Cout * 9
    r = []
    for big in ("0", "1"):
        os.environ["SST_CONV_BIG"] = big
        t = timeit(lambda: ops.conv_fwd(x, wp, Cout, 3, s))
        r.append(f"{t:7.1f} us {flops/t/1e6:6.1f} TF/s")
    tiles = B * ((ho + 7) // 8) * ((wo + 7) // 8) * ((Cout + 63) // 64)
    print(f"B{B} {H}x{W} {Cin}->{Cout} s{s} ({flops/1e9:5.2f} GF, {tiles} big tiles):  32x32 {r[0]}   64x64 {r[1]}")
del os.environ["SST_CONV_BIG"]

#!/usr/bin/env python3
"""Ablation timing of the conv forward kernel on the trunk shape (B=16, 24x24, 64->64).  Dev tool, GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops, _abi

def timeit(fn, n=50, reps=5):
    """GPU-bound time per launch: capture n launches in a hipGraph, replay (no host gaps)."""
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

shapes = [(16, 24, 24, 64, 64, 3, 1), (16, 48, 48, 64, 256, 3, 1), (16, 96, 96, 64, 3, 9, 1), (16, 96, 96, 3, 64, 9, 1)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for (B, H, W, Cin, Cout, k, s) in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda")
    flops = 2.0 * B * H * W * Cin * Cout * k * k
    print(f"shape B{B} {H}x{W} {Cin}->{Cout} k{k}: {flops/1e9:.2f} GFLOP")
    for name, dbg in [("full", 0), ("no-staging", 1), ("no-kloop", 2), ("no-epilogue", 4), ("staging only", 6), ("kloop only", 5), ("epilogue only", 3), ("empty", 7)]:
        t = timeit(lambda: ops.conv_fwd(x, wp, Cout, k, s, out_mode=dbg << 8))
        print(f"   {name:14s} {t:8.1f} us   {flops/t/1e6:7.1f} TFLOP/s-equivalent")
    for kb in (0, 24, 40, 60, 100):
        t = timeit(lambda: ops.conv_fwd(x, wp, Cout, k, s, out_mode=(kb << 4) << 8))
        print(f"   full, +{kb:3d} KB LDS pad   {t:8.1f} us   {flops/t/1e6:7.1f} TFLOP/s")
    t = timeit(lambda: ops.conv_fwd(x, wp, Cout, k, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
    print(f"   {'full+bn+stats':14s} {t:8.1f} us   {flops/t/1e6:7.1f} TFLOP/s")

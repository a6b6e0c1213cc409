#!/usr/bin/env python3
"""Dev tool: GPU-bound timing of conv_wgrad (trunk + up2 + deep-D shapes) with ablation bits ($SST_WGRAD_DBG)."""
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

shapes = [(16, 24, 24, 64, 64, 1), (16, 24, 24, 64, 256, 1), (16, 48, 48, 64, 256, 1), (16, 12, 12, 256, 512, 1), (16, 12, 12, 512, 512, 2)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, H, W, Cin, Cout, s) in shapes:
    x = torch.randn(B, H, W, Cin, device="cuda")
    ho, wo = ops.conv_out_hw(H, W, 3, s)
    dy = torch.randn(B, ho, wo, Cout, device="cuda")
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    flops = 2.0 * B * ho * wo * Cin * Cout * 9
    print(f"wgrad B{B} {H}x{W} {Cin}->{Cout} s{s}: {flops/1e9:.2f} GFLOP")
    variants = [("full", 0), ("no loads", 1), ("no mfma", 2), ("no store", 4), ("loads only", 6), ("mfma only", 5), ("empty", 7)]
    if os.environ.get("SST_ONLY_FULL"):
        variants = variants[:1]
    for name, dbg in variants:
        os.environ["SST_WGRAD_DBG"] = str(dbg)
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, s))
        print(f"   {name:12s} {t:8.1f} us  {flops/t/1e6:7.1f} TF/s-eq")
os.environ["SST_WGRAD_DBG"] = "0"

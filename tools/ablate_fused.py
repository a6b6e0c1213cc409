#!/usr/bin/env python3
"""Dev tool: trunk-shape conv variants, GPU-bound timing: plain / +bwd-stats epilogue / fully fused BN-backward stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops

def timeit(fn, n=50, reps=5):
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

B, H, W, C = 16, 24, 24, 64
x = torch.randn(B, H, W, C, device="cuda"); y2 = torch.randn(B, H, W, C, device="cuda"); res = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wd = ops.pack_conv(w, 1)
v = lambda: torch.rand(C, device="cuda") + 0.5
cA, cB, cC, sc, sh = v(), v(), v(), v(), v()
sl = torch.tensor([0.25], device="cuda")
print("plain conv              %.1f us" % timeit(lambda: ops.conv_fwd(x, wd, C, 3, 1)))
print("plain + residual        %.1f us" % timeit(lambda: ops.conv_fwd(x, wd, C, 3, 1, residual=res)))
print("fwd bn prologue + stats %.1f us" % timeit(lambda: ops.conv_fwd(x, wd, C, 3, 1, in_scale=sc, in_shift=sh, in_slope=sl, in_act=1, want_stats=True)))
print("dgrad + epi partials    %.1f us" % timeit(lambda: ops.conv_dgrad_bwdstats(x, wd, C, 3, y2, epi_scale=sc, epi_shift=sh, epi_slope=sl, epi_act=1)))
print("fused apply (no epi)    %.1f us" % timeit(lambda: ops.conv_dgrad_fused(x, y2, wd, C, 3, cA, cB, cC)))
print("fused apply+act+res+epi %.1f us" % timeit(lambda: ops.conv_dgrad_fused(x, y2, wd, C, 3, cA, cB, cC, in_scale=sc, in_shift=sh, in_slope=sl, in_act=1, residual=res, epi_y=y2)))
print("bwd_apply alone         %.1f us" % timeit(lambda: ops.bwd_apply(x, y2, scale=sc, shift=sh, slope=sl, act=1, cA=cA, cB=cB, cC=cC)))
part = ops.bwd_reduce(x, y2)
print("bwd_reduce alone        %.1f us" % timeit(lambda: ops.bwd_reduce(x, y2, scale=sc, shift=sh, slope=sl, act=1)))
m, r = v(), v()
dg, db, ds = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(1, device="cuda")
_, _, pt = ops.conv_dgrad_fused(x, y2, wd, C, 3, cA, cB, cC, epi_y=y2)
print("finalize2 (288 partials)%.1f us" % timeit(lambda: ops.bwd_finalize(pt, 9216, m, r, cA, dg, db, ds)))
_, _, st, cnt = ops.conv_fwd(x, wd, C, 3, 1, want_stats=True)
gm, bt, rm, rv = v(), v(), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
print("bn_finalize (288 tiles) %.1f us" % timeit(lambda: ops.bn_finalize(st, cnt, gm, bt, rm, rv)))
print("bn_residual             %.1f us" % timeit(lambda: ops.bn_residual(x, sc, sh, y2)))

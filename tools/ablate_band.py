#!/usr/bin/env python3
"""Ablation timing of the band conv kernel (csrc/conv_band.hip) vs the general kernel on the trunk shape.  Dev tool, GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops
from ablate_conv import timeit  # noqa: E402  (same directory)

B, H, W, C = 16, 24, 24, 64
if len(sys.argv) > 2:
    B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[2])
x = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wp, wd = ops.pack_conv(w), ops.pack_conv(w, 1)
sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
y2, res, ys = (torch.randn(B, H, W, C, device="cuda") for _ in range(3))
cA, cB, cC = (torch.randn(C, device="cuda") for _ in range(3))
flops = 2.0 * B * H * W * C * C * 9
for band in ("3", "9", "0"):
    os.environ["SST_CONV_BAND"] = band
    print(f"SST_CONV_BAND={band}  B{B} {H}x{W}")
    for name, dbg in [("full", 0), ("no-staging", 1), ("no-kloop", 2), ("no-epilogue", 4), ("staging only", 6), ("kloop only", 5),
                      ("epilogue only", 3), ("empty", 7)]:
        t = timeit(lambda: ops.conv_fwd(x, wp, C, 3, 1, out_mode=dbg << 8))
        print(f"   {name:14s} {t:8.1f} us   {flops/t/1e6:7.1f} TF/s-equiv")
    t = timeit(lambda: ops.conv_fwd(x, wp, C, 3, 1, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
    print(f"   {'fwd+bn+stats':14s} {t:8.1f} us   {flops/t/1e6:7.1f} TF/s")
    t = timeit(lambda: ops.conv_dgrad_fused(x, y2, wd, C, 3, cA=cA, cB=cB, cC=cC, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1,
                                            residual=res, epi_y=ys, epi_scale=sc, epi_shift=sh, epi_slope_const=0.2, epi_act=1))
    print(f"   {'fused bwd stage':14s} {t:8.1f} us   {flops/t/1e6:7.1f} TF/s")

# ---- accumulator mode (fp64 atomics in the epilogue, BN affine from accumulators in the prologue)
os.environ["SST_CONV_BAND"] = "3"
acc_in = torch.rand(ops.ACC_NREP, C, 2, device="cuda", dtype=torch.float64) + 1.0
acc_in[..., 1] += 50.0
acc_out = torch.zeros(ops.ACC_NREP, C, 2, device="cuda", dtype=torch.float64)
gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
outs = tuple(torch.empty(C, device="cuda") for _ in range(4))
n = float(B * H * W)
print("accumulator mode, NB=3")
print("   plain fwd (reference point)  %.1f us" % timeit(lambda: ops.conv_fwd(x, wp, C, 3, 1)))
print("   fwd + bn prologue + stats    %.1f us" % timeit(lambda: ops.conv_fwd(x, wp, C, 3, 1, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True)))
print("   producer only (atomics)      %.1f us" % timeit(lambda: ops.conv_fwd_acc(x, wp, C, 3, in_slope_const=0.2, in_act=1, st_acc=acc_out)))
print("   consumer only (acc prologue) %.1f us" % timeit(lambda: ops.conv_fwd_acc(x, wp, C, 3, in_slope_const=0.2, in_act=1, in_acc=acc_in, in_bn=(gam, bet), n=n, out_stats=outs)))
print("   both                         %.1f us" % timeit(lambda: ops.conv_fwd_acc(x, wp, C, 3, in_slope_const=0.2, in_act=1, in_acc=acc_in, in_bn=(gam, bet), n=n, out_stats=outs, st_acc=acc_out)))
bacc_in = torch.rand(ops.ACC_NREP, C, 4, device="cuda", dtype=torch.float64)
bacc_out = torch.zeros(ops.ACC_NREP, C, 4, device="cuda", dtype=torch.float64)
mean_, rstd_ = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5
dg_, db_, ds_ = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(1, device="cuda")
print("backward stage, NB=3")
print("   plain dgrad                          %.1f us" % timeit(lambda: ops.conv_fwd(x, wd, C, 3, 1)))
print("   + fused input (y2, act, cA..) + dy   %.1f us" % timeit(lambda: ops.conv_dgrad_fused(x, y2, wd, C, 3, cA=cA, cB=cB, cC=cC, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1)))
print("   + residual + epilogue partial tiles  %.1f us" % timeit(lambda: ops.conv_dgrad_fused(x, y2, wd, C, 3, cA=cA, cB=cB, cC=cC, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, residual=res, epi_y=ys, epi_scale=sc, epi_shift=sh, epi_slope_const=0.2, epi_act=1)))
print("   accumulator mode (in + out)          %.1f us" % timeit(lambda: ops.conv_dgrad_fused_acc(x, wd, C, 3, y2=y2, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, residual=res, epi_y=ys, epi_scale=sc, epi_shift=sh, epi_slope_const=0.2, epi_act=1, bw_in_acc=bacc_in, bn=(mean_, rstd_, gam), n=n, dgamma=dg_, dbeta=db_, dslope=ds_, bw_st_acc=bacc_out)))

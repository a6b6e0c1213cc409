#!/usr/bin/env python3
"""Dev tool: the split-operand bf16 MFMA proposal, measured - (a) accuracy of C = A B (32 x K x 32, conv-like
operand statistics) through the fp32 MFMA and through six bf16 MFMAs per 16 k, both against fp64; (b) the rate of the six-MFMA group
next to the fp32 MFMA's (tools/mfma_peak.py) in register-only loops."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import _abi
lib = _abi.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
for K in (64, 576, 4608):
    A = torch.randn(32, K, generator=g)
    B = torch.randn(K, 32, generator=g) / K ** 0.5
    ref = A.double() @ B.double()
    Ad, Bd = A.cuda(), B.cuda()
    c32, c3 = torch.zeros(32, 32, device="cuda"), torch.zeros(32, 32, device="cuda")
    _abi.check(lib.sst_debug_bf16x3(Ad.data_ptr(), Bd.data_ptr(), c32.data_ptr(), c3.data_ptr(), K, 0, 1, s), "probe")
    torch.cuda.synchronize()
    e = lambda c: float((c.cpu().double() - ref).norm() / ref.norm())
    m = lambda c: float(((c.cpu().double() - ref).abs() / ref.abs().clamp_min(1e-3)).max())
    t32 = (A @ B)
    print(f"K {K:5d}: norm-wise error vs fp64  fp32 MFMA {e(c32):.2e}  six bf16 MFMAs {e(c3):.2e}  (torch CPU fp32 {e(t32.cuda()):.2e});"
          f"  worst element  {m(c32):.2e} / {m(c3):.2e}", flush=True)
out = torch.zeros(16, device="cuda")
for blocks in (256, 1280):
    for iters in (2000, 20000):
        res = {}
        for name, fn, flop in (("fp32 MFMA", lambda: lib.sst_debug_mfma_peak(out.data_ptr(), blocks, iters, s), 4 * 32 * 32 * 2 * 2.0),
                               ("six bf16 MFMAs per 16 k", lambda: lib.sst_debug_bf16x3(0, 0, out.data_ptr(), 0, iters, 1, blocks, s), 4 * 32 * 32 * 16 * 2.0)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5 * 1e-3
            res[name] = blocks * 4 * iters * flop / t / 1e12
        print(f"blocks {blocks:5d} iters {iters:6d}: " + "   ".join(f"{k}: {v:7.1f} fp32-equivalent TFLOP/s" for k, v in res.items()), flush=True)

#!/usr/bin/env python3
"""Dev tool: sustained fp32-MFMA rate of the chip (no memory traffic) for launches of different length / occupancy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import _abi
lib = _abi.lib()
out = torch.zeros(16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 1280):
    for iters in (200, 2000, 20000):
        n = 20 if iters < 20000 else 5
        for _ in range(3):
            lib.sst_debug_mfma_peak(out.data_ptr(), blocks, iters, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(n):
            lib.sst_debug_mfma_peak(out.data_ptr(), blocks, iters, s)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        fl = blocks * 4 * iters * 4 * 32 * 32 * 2 * 2.0
        print(f"blocks {blocks:5d} iters {iters:6d}: {t*1e6:9.1f} us/launch  {fl/t/1e12:6.1f} TFLOP/s", flush=True)

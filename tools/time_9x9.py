#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the three 9x9 kernels with a 3-channel side at the bench size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from srganst import ops
from ablate_wgrad import timeit  # noqa

B, H, C = 16, 96, 64
u = torch.randn(B, H, H, C, device="cuda")
w3 = torch.randn(3, C, 9, 9, device="cuda") * 0.02
b3 = torch.zeros(3, device="cuda")
g3 = torch.randn(B, H, H, 3, device="cuda")
dw3 = torch.empty(3, C, 9, 9, device="cuda")
fl = 2.0 * B * H * H * C * 3 * 81
t = timeit(lambda: ops.conv9_to3_fwd(u, w3, bias=b3, in_slope_const=0.25, in_act=1, want_pre=True))
print(f"conv3 fwd  (to3)      {t:7.1f} us  {fl/t/1e6:6.1f} TF/s")
t = timeit(lambda: ops.conv9_c3_fwd(g3, w3, 1))
print(f"conv3 dgrad (c3 fwd)  {t:7.1f} us  {fl/t/1e6:6.1f} TF/s")
t = timeit(lambda: ops.wgrad_c3(u, g3, dw3, 0, in_slope_const=0.25, in_act=1))
print(f"conv3 wgrad (c3)      {t:7.1f} us  {fl/t/1e6:6.1f} TF/s")

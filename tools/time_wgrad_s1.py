#!/usr/bin/env python3
"""Dev tool: GPU-bound time of the stride-1 weight gradients of the discriminator / up-sampler shapes (B = 16), incl. the slab reduce."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import ops, _abi
from time_pipe_lib import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, cin, cout) in [(48, 64, 128), (24, 128, 256), (12, 256, 512), (24, 64, 256), (48, 64, 256)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    dy = torch.randn(B, H, H, cout, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    fl = 2.0 * B * H * H * cin * cout * 9
    name = _abi.lib().sst_conv_wgrad_kernel_name(B, H, H, cin, cout, 3, 1, 1).decode()
    nch = _abi.lib().sst_conv_wgrad_chunks2(B, H, H, cin, cout, 3, 1, 1)
    t = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 1, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1))
    os.environ["SST_WGRAD_S1T"] = "0"
    t0 = timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 1, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1))
    os.environ.pop("SST_WGRAD_S1T")
    print(f"B{B} {H:3d}px {cin:3d}->{cout:3d} s1 ({fl/1e9:5.2f} GF, ideal {fl/157.3e6:5.1f} us): {t:6.1f} us {fl/t/1e6:5.1f} TF  {name} chunks {nch} slab {nch*9*cin*cout*4/1e6:.1f} MB | band kernel {t0:6.1f} us", flush=True)

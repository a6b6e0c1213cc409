#!/usr/bin/env python3
"""Dev tool: phase ablation of the pipelined conv kernel on the ablation build of the library (tools/build_ablate.sh ->
build_ab/libsrganst.so, -DSST_PIPE_ABLATE).  SST_PIPE_DBG bits: 1 no LDS staging writes, 2 no epilogue, 4 no patch loads, 8 weight
refills from one cache-resident address, 16 no MFMAs.  usage: SST_LIB_PATH=build_ab/libsrganst.so python tools/ablate_pipe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SST_LIB_PATH", os.path.join(ROOT, "build_ab", "libsrganst.so"))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from srganst import _abi, ops
from time_pipe_lib import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VARIANTS = [("full", 0), ("-store", 1), ("-epi", 2), ("-patch ld", 4), ("-w ld", 8), ("-store-epi", 3), ("mfma only", 15), ("no mfma", 16), ("nothing", 31)]
print("variants: " + ", ".join(f"{n}={d}" for n, d in VARIANTS))
for (H, cin, cout, s) in [(96, 64, 64, 2), (48, 128, 128, 2), (24, 256, 256, 2), (12, 512, 512, 2), (48, 64, 128, 1), (24, 128, 256, 1)]:
    x = torch.randn(B, H, H, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wp = ops.pack_conv(w)
    ho = (H - 1) // s + 1
    fl = 2.0 * B * ho * ho * cin * cout * 9
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    row = f"B{B} {H:3d}px {cin:3d}->{cout:3d} s{s} (157 TF: {fl/157.3e6:5.1f} us):"
    for name, dbg in VARIANTS:
        os.environ["SST_PIPE_DBG"] = str(dbg)
        _abi.reload_env()
        t = timeit(lambda: ops.conv_fwd(x, wp, cout, 3, s, in_scale=sc, in_shift=sh, in_slope_const=0.2, in_act=1, want_stats=True))
        row += f" | {name} {t:6.1f}"
    os.environ["SST_PIPE_DBG"] = "0"
    _abi.reload_env()
    print(row, flush=True)

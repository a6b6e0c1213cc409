#!/usr/bin/env python3
"""Dev tool: run the bench step for many iterations on a fixed synthetic batch and print the losses (sanity: finite, decreasing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd")); sys.path.insert(0, ROOT)
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "srresnet"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
eng, cfg = bench.build_engine(wl, "cuda", use_graph=True, hr=96)
gt, lr = bench.synth_batch(16, 96, "cuda", 1)
for i in range(n + 1):
    out = eng.step(gt, lr)
    if i % (n // 6) == 0:
        vals = out[0] if isinstance(out, tuple) else out
        v = {k: round(float(x), 5) for k, x in vals.items()}
        ok = all(torch.isfinite(p).all().item() for p in eng.G.parameters())
        print(i, v, "params finite:", ok, flush=True)

#!/usr/bin/env python3
"""Dev tool: list the small aten kernels (fills, copies, adds) one eager training step issues, with the Python frames that issue them."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "srresnet"
dev = torch.device("cuda:0")
eng, cfg = bench.build_engine(wl, dev, use_graph=False, hr=96)
gt, lr = bench.synth_batch(16, 96, dev, 0)
for _ in range(3):
    eng.step(gt, lr)
torch.cuda.synchronize()
cnt = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ("view", "empty", "as_strided", "detach", "alias", "reshape", "_unsafe_view", "select", "slice", "expand", "permute", "t.default", "transpose", "unsqueeze", "squeeze", "_local_scalar")):
            fr = [f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in traceback.extract_stack() if "srganst" in f.filename]
            cnt[(name, tuple(fr[-3:]))] += 1
        return func(*args, **(kwargs or {}))


with Log():
    eng.step(gt, lr)
torch.cuda.synchronize()
for (name, st), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(n, name, " <- ".join(reversed(st)))

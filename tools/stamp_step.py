#!/usr/bin/env python3
"""Dev tool: when do the two branches of the captured iteration (discriminator step on the side stream, generator backward on the
main stream) start and end in an UNPROFILED hipGraph replay?  Device wall-clock stamps (sst_debug_stamp) at the marked points of
engine.TrainEngine._iter_gd, read back after a few replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "srgan-st_amd"))
os.environ["SST_STAMP"] = "1"
import torch
import bench
from srganst import ops
dev = torch.device("cuda:0")
eng, cfg = bench.build_engine("srgan", dev, os.environ.get("SST_EAGER", "0") == "0", 96)
gt, lr = bench.synth_batch(16, 96, dev, 0)
names = ["start", "G fwd done", "D(sr)+losses done (fork)", "side: start", "side: D fwd x2 + cls bwd done", "side: D feature bwd done",
         "main: G bwd start", "main: G bwd done", "main: G Adam done", "end (D Adam done)"]
for it in range(8):
    eng.step(gt, lr)
    torch.cuda.synchronize()
    if it >= 5:
        t = ops.debug_stamps().cpu().tolist()
        print(f"-- replay {it}")
        for i, n in enumerate(names):
            print(f"  {(t[i] - t[0]) / 100.0:9.1f} us  {n}")

#!/usr/bin/env python3
"""Dev tool: dump the captured iteration graph of the bench workload (hipGraphDebugDotPrint) into gpurun_out/graph_dot/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "graph_dot")
os.makedirs(out, exist_ok=True)
os.environ["SST_GRAPH_DOT"] = out
import torch
import bench
dev = torch.device("cuda:0")
eng, cfg = bench.build_engine("srgan", dev, True, 96)
gt, lr = bench.synth_batch(16, 96, dev, 0)
for _ in range(5):
    eng.step(gt, lr)
torch.cuda.synchronize()
print(os.listdir(out))

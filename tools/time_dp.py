#!/usr/bin/env python3
"""Dev tool: iteration time of the data-parallel schedules on ONE GPU (RCCL world 1, collectives really issued):
blocking (round 1), overlapped + two-branch (default for N > 1), and the single-process merged graph for reference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "srgan-st_amd")]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as td
import bench
from srganst.config import Config
from srganst.engine import TrainEngine
from srganst.loss import MSELoss, StructureTensorLoss
from srganst.model import Discriminator, Generator

torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
gt, lr = bench.synth_batch(16, 96, dev, 1)
for name, kw in (("single process, merged graph", dict()), ("dp blocking", dict(force_dp=True, overlap_comm=False)),
                 ("dp overlapped, graphs cut at the collectives", dict(force_dp=True, overlap_comm=True, one_graph=False)),
                 ("dp, ONE graph with captured collectives", dict(force_dp=True, overlap_comm=True))):
    cfg = Config()
    cfg.DIST.ONE_GRAPH = kw.pop("one_graph", True)
    cfg.KERNEL.REUSE_D_SR = os.environ.get("SST_REUSE_D_SR", "0") != "0"       # default here: three discriminator forwards, as the bench headline
    torch.manual_seed(0)
    D, G = Discriminator(cfg).to(dev).train(), Generator(cfg).to(dev).train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=True, **kw)
    for _ in range(8):
        eng.step(gt, lr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        eng.step(eng.gt, eng.lr)
    torch.cuda.synchronize()
    print(f"{name:32s} {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms / iteration  (graphs active: {eng.graph_active})", flush=True)
    if os.environ.get("SST_STAMP", "0") != "0":
        from srganst import ops
        t = ops.debug_stamps().cpu().tolist()
        print("   stamps (us): " + "  ".join(f"{i}:{(t[i] - t[0]) / 100.0:.0f}" for i in range(10)), flush=True)
    eng.close()
td.destroy_process_group()

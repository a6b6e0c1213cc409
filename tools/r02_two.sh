#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-two}
cd $R
cat > /tmp/two_check.py <<'PY'
import os, sys
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "srgan-st_amd")]
import torch
from srganst.config import Config
from srganst.engine import TrainEngine
from srganst.loss import MSELoss, StructureTensorLoss
from srganst.model import Discriminator, Generator

def run(two, graph, gd=False):
    cfg = Config(); cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB, cfg.MODEL.D_N_CHANNEL = 16, 2, 16
    cfg.KERNEL.D_TWO_STREAMS = two
    cfg.KERNEL.OVERLAP_GD = gd
    torch.manual_seed(1)
    D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0); cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=graph, adam_capturable=True)
    gen = torch.Generator().manual_seed(2)
    for _ in range(5):
        eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
    torch.cuda.synchronize()
    assert eng.graph_active == graph
    sd = {"G." + k: v.clone() for k, v in G.state_dict().items()}
    sd.update({"D." + k: v.clone() for k, v in D.state_dict().items()})
    return sd
a = run(False, True)
for two, graph, gd in ((True, False, False), (True, True, False), (True, False, True), (True, True, True)):
    b = run(two, graph, gd)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print("two-stream", two, "graph", graph, "overlap G/D", gd, "mismatching tensors:", bad[:5], len(bad))
PY
timeout -k 10 300 python3 /tmp/two_check.py 2>&1 | tail -5
for v in 0 1 0 1; do echo -n "SST_D_TWO_STREAMS=1 SST_OVERLAP_GD=$v: "; SST_OVERLAP_GD=$v SST_D_TWO_STREAMS=1 timeout -k 10 300 python3 bench.py --steps 100 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['hip_graph'])"; done

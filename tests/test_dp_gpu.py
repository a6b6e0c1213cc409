"""GPU, 2 ranks sharing the one test GPU (gloo carries the collective; RCCL refuses two ranks on one device):
the data-parallel engine path - [fwd+bwd graph] -> gradient all-reduce -> [optimizer graph] - end to end.
Parity statement: after each step every rank holds the parameters a single process would get from
the MEAN of the per-rank gradients; BatchNorm statistics stay rank-local."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(seed=3):
    from srganst.config import Config
    from srganst.model import Generator
    cfg = Config()
    cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB = 16, 2
    torch.manual_seed(seed)
    return cfg, Generator(cfg).cuda().train()


def _batch(rank, step):
    g = torch.Generator().manual_seed(1000 * rank + step)
    return torch.rand(2, 3, 32, 32, generator=g).cuda(), torch.rand(2, 3, 8, 8, generator=g).cuda()


def _worker(rank, world, port, use_graph, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "srgan-st_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from srganst import dist as sdist
    from srganst.engine import WarmupEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    sdist.init_from_env("gloo")
    cfg, G = _make(seed=3 + rank)                       # different init per rank ...
    sdist.broadcast_module(G)                           # ... made identical
    eng = WarmupEngine(cfg, G, {"Pixel": MSELoss(), "ST": StructureTensorLoss()}, {"Pixel": 1.0, "ST": 1 / 3},
                       use_graph=use_graph, adam_capturable=True)
    assert eng.world == 2
    for step in range(4):
        eng.step(*_batch(rank, step))
    torch.cuda.synchronize()
    q.put((rank, {k: v.cpu().numpy() for k, v in G.state_dict().items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_dp_world2_matches_mean_gradient_step(use_graph):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in res.items()}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # parameters identical on both ranks; BN running stats are rank-local (different data)
    for k in res[0]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(res[0][k], res[1][k]), k
    assert not torch.equal(res[0]["trunk.0.rcb.1.running_mean"], res[1]["trunk.0.rcb.1.running_mean"])

    # single-process emulation: per-rank forward/backward on a replica carrying that rank's BN buffers, mean gradient,
    # one Adam on the shared parameters
    from srganst.engine import make_adam
    from srganst.loss import MSELoss, StructureTensorLoss
    cfg, G0 = _make(seed=3)                             # rank 0's init == the broadcast state
    _, G1 = _make(seed=3)
    G1.load_state_dict(G0.state_dict())
    opt = make_adam(G0, 1e-4, (0.9, 0.999), 1e-4, 0, capturable=True)      # non-flat grads below: exercises the stock-Adam fallback
    mse, st = MSELoss(), StructureTensorLoss()
    for step in range(4):
        grads = []
        for rank, G in enumerate((G0, G1)):
            G.zero_grad()
            gt, lr = _batch(rank, step)
            sr = G(lr)
            (mse(sr, gt) + st(sr, gt) * (1 / 3)).backward()
            grads.append([p.grad.clone() for p in G.parameters()])
        for p, a, b in zip(G0.parameters(), *grads):
            p.grad = (a + b) * 0.5
        opt.step()
        with torch.no_grad():
            for p0, p1 in zip(G0.parameters(), G1.parameters()):
                p1.copy_(p0)
    for k, v in G0.state_dict().items():
        if "num_batches" in k:
            continue
        assert torch.allclose(v.cpu(), res[0][k], rtol=1e-5, atol=1e-7), k
    for k, v in G1.state_dict().items():
        if "running" in k:
            assert torch.allclose(v.cpu(), res[1][k], rtol=1e-5, atol=1e-7), k


def _nccl_worker(port, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "srgan-st_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as td
    from srganst.engine import WarmupEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))    # RCCL communicator + watchdog thread
    out = []
    for force_dp in (False, True):
        cfg, G = _make(seed=5)
        eng = WarmupEngine(cfg, G, {"Pixel": MSELoss(), "ST": StructureTensorLoss()}, {"Pixel": 1.0, "ST": 1 / 3},
                           use_graph=True, adam_capturable=True, force_dp=force_dp)
        for step in range(5):
            eng.step(*_batch(0, step))
        torch.cuda.synchronize()
        assert eng._fb.graph is not None, "hipGraph capture fell back to eager next to a live RCCL communicator"
        out.append({k: v.cpu().numpy() for k, v in G.state_dict().items()})
        eng.close()                 # release this engine's graphs now (not by a cyclic collection during a later capture)
    # the full G + D step (train.py:100-164): [G fwd+bwd graph] -> all-reduce -> [G Adam] -> [D fwd+bwd graph] -> 2-bucket all-reduce -> [D Adam]
    from srganst.engine import TrainEngine
    from srganst.model import Discriminator
    # single graphs / split graphs with blocking collectives (round-1 schedule) / overlapped schedule with the graphs cut where a
    # collective goes out (the two D buckets and G's message in flight on RCCL's stream while the next graphs run) / the default with
    # RCCL: ONE graph per iteration with the three collectives captured inside it (DIST.ONE_GRAPH)
    # ... and the last three again with 8 images per batch: the discriminator step's two passes then run as ONE batch (coefficient
    # groups) on every one of them
    for force_dp, overlap, one_graph, nb in ((False, False, True, 2), (True, False, True, 2), (True, True, False, 2), (True, True, True, 2),
                                             (False, False, True, 8), (True, True, False, 8), (True, True, True, 8)):
        cfg, G = _make(seed=6)
        cfg.DIST.ONE_GRAPH = one_graph
        cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
        cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
        cfg.SOLVER.D_UPDATE_INTERVAL = 1
        torch.manual_seed(7)
        D = Discriminator(cfg).to("cuda:0").train()
        eng = TrainEngine(cfg, G, D, use_graph=True, adam_capturable=True, force_dp=force_dp, overlap_comm=overlap)
        assert eng.one_graph_dp == (force_dp and overlap and one_graph) and eng.overlap == (overlap and not eng.one_graph_dp)
        for step in range(4):
            g = torch.Generator().manual_seed(50 + step)           # the discriminator is fixed to 96 x 96 inputs (model.py:31-34)
            eng.step(torch.rand(nb, 3, 96, 96, generator=g).cuda(), torch.rand(nb, 3, 24, 24, generator=g).cuda())
        torch.cuda.synchronize()
        assert eng.graph_active, "a hipGraph capture fell back to eager next to a live RCCL communicator"
        assert eng.d_batched == (nb == 8)
        sd = {"G." + k: v.cpu().numpy() for k, v in G.state_dict().items()}
        sd.update({"D." + k: v.cpu().numpy() for k, v in D.state_dict().items()})
        out.append(sd)
        eng.close()
    q.put(out)
    td.barrier()
    td.destroy_process_group()


def test_graph_capture_next_to_rccl_communicator():
    """RCCL (backend nccl) initialised, its watchdog thread alive: hipGraph capture must still work, and the split
    [fwd+bwd graph] -> all-reduce over RCCL -> [optimizer graph] path must equal the single-graph path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    a, b, c, d, e, f, c8, e8, f8 = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    for k in c8:
        assert (c8[k] == e8[k]).all() and (c8[k] == f8[k]).all(), k      # the batched discriminator step under the collectives
    for k in a:
        assert (a[k] == b[k]).all(), k           # SRResNet step: all-reduce captured inside the step's graph == single-process graph
    for k in c:
        assert (c[k] == d[k]).all(), k
        assert (c[k] == e[k]).all(), k           # overlapped collectives between cut graphs: bit-identical to the single-graph step
        assert (c[k] == f[k]).all(), k           # collectives captured inside the ONE iteration graph: bit-identical too


def _train_worker(rank, world, port, overlap, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "srgan-st_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from srganst import dist as sdist
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator
    sdist.init_from_env("gloo")
    cfg, G = _make(seed=3 + rank)
    cfg.MODEL.D_N_CHANNEL = 8
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    torch.manual_seed(11 + rank)
    D = Discriminator(cfg).cuda().train()
    sdist.broadcast_module(G)
    sdist.broadcast_module(D)
    eng = TrainEngine(cfg, G, D, use_graph=True, adam_capturable=True, overlap_comm=overlap)
    assert eng.world == 2 and eng.overlap == overlap
    for step in range(4):
        g = torch.Generator().manual_seed(1000 * rank + step)
        eng.step(torch.rand(2, 3, 96, 96, generator=g).cuda(), torch.rand(2, 3, 24, 24, generator=g).cuda())
    torch.cuda.synchronize()
    assert eng.graph_active
    sd = {"G." + k: v.cpu().numpy() for k, v in G.state_dict().items()}
    sd.update({"D." + k: v.cpu().numpy() for k, v in D.state_dict().items()})
    q.put((rank, sd))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp_world2_overlapped_schedule_is_bit_identical():
    """Full G + D iteration on 2 ranks (gloo carries the collectives): the overlapped schedule - G's message under the D forward,
    classifier bucket under the feature-stack backward, D half without autograd - gives bit for bit the parameters of the blocking
    schedule, identical on both ranks; BatchNorm buffers stay rank-local."""
    ctx = mp.get_context("spawn")
    res = {}
    for overlap in (False, True):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
        for p in procs:
            p.start()
        out = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        res[overlap] = out
    for k in res[True][0]:
        for r in (0, 1):
            assert (res[True][r][k] == res[False][r][k]).all(), (k, r)
        if "running" not in k and "num_batches" not in k:
            assert (res[True][0][k] == res[True][1][k]).all(), k
    assert not (res[True][0]["D.features.3.running_mean"] == res[True][1]["D.features.3.running_mean"]).all()


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the form the driver's N = 1 command has): the parent starts two rank
    processes itself (gloo carries the collectives: both ranks share the one test GPU) and relays rank 0's line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "4",
                        "--no-roofline", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["comm"] == {"backend": "gloo", "ranks": 2}
    assert out["config"]["global_batch"] == 32 and out["config"]["hip_graph"] is True

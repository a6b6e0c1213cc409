"""GPU parity of the discriminator path + the full SRGAN iteration against the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_fp64_truth, oracle_grads, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def ops():
    from srganst import ops
    return ops


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64), (1, 13, 9, 8, 16), (2, 12, 12, 128, 128), (1, 6, 6, 64, 32), (1, 7, 10, 64, 64),
                                  (1, 10, 18, 40, 24), (3, 16, 8, 96, 72), (2, 96, 96, 64, 64)])
def test_conv_s2_dgrad(ops, case):
    # even H and W with channel counts that are multiples of 4 take the merged-classes kernel (conv_s2dgrad4_kernel: partial
    # tiles, partial channel blocks on either side); odd sizes the per-class launch
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y = F.conv2d(x, w.double(), None, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dx = ops.conv_s2_dgrad(nhwc(dy).cuda(), ops.pack_conv_s2_dgrad(w.cuda()), H, W, Cin)
    assert rel_err(nchw(dx.cpu()), x.grad) < TOL


@pytest.mark.parametrize("case", [(16, 1024, 18432), (32, 1024, 18432), (4, 40, 576), (16, 1024, 1000), (33, 24, 100), (33, 64, 96), (1, 128, 64)])
def test_linear_fwd_dgrad_wgrad(ops, case):
    # (the weight gradient runs on the matrix cores when K % 32 == 0 and N % 64 == 0 - odd batch sizes included - else as FMAs)
    M, N, K = case
    g = torch.Generator().manual_seed(22)
    x = torch.randn(M, K, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(N, K, generator=g, dtype=torch.float64) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.linear(x, w, b)
    dy = torch.randn(M, N, generator=g)
    y.backward(dy.double())
    xd, wd, bd, dyd = x.detach().float().cuda(), w.detach().float().cuda(), b.detach().float().cuda(), dy.cuda()
    assert rel_err(ops.linear_fwd(xd, wd, bd).cpu(), y.detach()) < TOL
    assert rel_err(ops.linear_dgrad(dyd, wd).cpu(), x.grad) < TOL
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    ops.linear_wgrad(dyd, xd, dw, db)
    assert rel_err(dw.cpu(), w.grad) < TOL and rel_err(db.cpu(), b.grad) < TOL
    ops.linear_wgrad(dyd, xd, dw, db, accumulate=True)                  # the second pass of a two-pass step adds into the first one's
    assert rel_err(dw.cpu(), 2 * w.grad) < TOL and rel_err(db.cpu(), 2 * b.grad) < TOL


def test_linear_dgrad_nhwc_scatter_and_flatten(ops):
    B, C, H, W, N = 3, 8, 3, 5, 24
    g = torch.Generator().manual_seed(23)
    y = torch.randn(B, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.leaky_relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2).flatten(1)
    flat = ops.flatten_act(nhwc(y).cuda(), sc.cuda(), sh.cuda(), 0.2, 1)
    assert torch.allclose(flat.cpu(), ref, rtol=1e-6, atol=1e-6)
    w = torch.randn(N, C * H * W, generator=g)
    dy = torch.randn(B, N, generator=g)
    dflat = dy.double() @ w.double()
    dx = ops.linear_dgrad(dy.cuda(), w.cuda(), nhwc=(C, H * W)).view(B, H, W, C)
    assert rel_err(nchw(dx.cpu()).flatten(1), dflat) < TOL


def test_head(ops):
    g = torch.Generator().manual_seed(24)
    h = torch.randn(16, 1024, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(1, 1024, generator=g, dtype=torch.float64) / 32).requires_grad_(True)
    b = torch.randn(1, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.linear(F.leaky_relu(h, 0.2), w, b)
    dy = torch.randn(16, 1, generator=g)
    y.backward(dy.double())
    hd, wd, bd = h.detach().float().cuda(), w.detach().float().cuda(), b.detach().float().cuda()
    assert rel_err(ops.head_fwd(hd, wd, bd, 0.2).cpu(), y.detach()) < TOL
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    dh = ops.head_bwd(hd, wd, dy.cuda(), 0.2, dw, db)
    assert rel_err(dh.cpu(), h.grad) < TOL and rel_err(dw.cpu(), w.grad) < TOL and rel_err(db.cpu(), b.grad) < TOL


def test_bce_logits(ops):
    x = torch.randn(16, 1, dtype=torch.float64, requires_grad=True)
    for t in (0.9, 0.0):
        x.grad = None
        l = F.binary_cross_entropy_with_logits(x, torch.full_like(x, t))
        (l * 0.5).backward()
        loss, dl = ops.bce_logits(x.detach().float().cuda(), t, True, True, scale_host=0.5)
        assert abs(loss.item() - l.item()) < 1e-6 and rel_err(dl.cpu(), x.grad) < 1e-5


def make_cfg(ch=64, rcb=16, dch=64):
    from srganst.config import Config
    cfg = Config()
    cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB, cfg.MODEL.D_N_CHANNEL = ch, rcb, dch
    return cfg


def test_discriminator_full_seed0(golden):
    from srganst.model import Discriminator
    from srganst.loss import BCEWithLogitsLoss
    g = golden("d_full_seed0")
    torch.manual_seed(0)
    D = Discriminator(make_cfg())
    assert sum(p.numel() for p in D.parameters()) == 23563649            # reference model.py:194
    assert torch.equal(D.state_dict()["features.0.weight"], T(g["w/features.0.weight"]))
    sd0 = {k: v.clone() for k, v in D.state_dict().items()}
    D.cuda().train()
    x = T(g["x"]).cuda().requires_grad_(True)
    logit = D(x)
    assert torch.allclose(logit.detach().cpu(), T(g["logit"]), rtol=1e-3, atol=1e-4)
    loss = BCEWithLogitsLoss()(logit, torch.full([2, 1], 0.9).cuda())
    assert abs(loss.item() - g["loss"].item()) < 1e-4
    loss.backward()
    # gradients: fp64-truth criterion (conftest.assert_fp64_truth): truth = the oracle in fp64 on the same weights / input; the
    # fp32 side of the bound is the REFERENCE's own fp32 gradient where the fixture stores it, the oracle's fp32 run elsewhere
    from oracle import model as om

    def fl(sdx, x_):
        lg = om.discriminator_forward(sdx, x_, True, {})
        return F.binary_cross_entropy_with_logits(lg, torch.full_like(lg, 0.9))
    ins = ((T(g["x"]), True),)
    _, g32, (dx32,), _ = oracle_grads(fl, sd0, torch.float32, ins)
    _, g64, (dx64,), _ = oracle_grads(fl, sd0, torch.float64, ins)
    assert_fp64_truth("dx", x.grad.cpu(), T(g["dx"]), dx64)
    named = dict(D.named_parameters())
    report = []
    for k, v in named.items():
        ref32 = T(g["g/" + k]) if ("g/" + k) in g.files else g32[k]
        assert_fp64_truth(k, v.grad.cpu(), ref32, g64[k], report)
    for f in g.files:
        if f.startswith("g/") and f.endswith("#head4"):
            k = f[2:-6]
            assert_fp64_truth(f, named[k].grad[:4].cpu(), T(g[f]), g64[k][:4])
    worst = max(report, key=lambda r: r[1])
    print(f"worst param-grad error vs fp64: {worst[0]} {worst[1]:.2e} (oracle fp32: {worst[2]:.2e})")


def test_gan_iteration_small_golden(golden):
    """One full train.py iteration through TrainEngine (eager): losses, all grads via Adam-updated states, D BN counters."""
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    g = golden("gan_small_iter")
    cfg = make_cfg(8, 2, 4)
    torch.manual_seed(0)
    D = Discriminator(cfg)                       # same construction order as the fixture: D then G
    G = Generator(cfg)
    for k, v in D.state_dict().items():
        key = "d_state0/" + k
        if key in g.files:
            assert torch.equal(v, T(g[key])), k
    d_state0 = {k: v.clone() for k, v in D.state_dict().items()}
    g_state0 = {k: v.clone() for k, v in G.state_dict().items()}
    D.cuda().train()
    G.cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=False)
    losses, d_loss = eng.step(T(g["gt"]).cuda(), T(g["lr"]).cuda())
    assert rel_err(eng.sr.cpu(), g["sr"]) < 1e-4
    for name in ("Adversarial", "Pixel", "ST"):
        assert abs(losses[name].item() - g["g_loss/" + name].item()) < 1e-3 * abs(g["g_loss/" + name].item()), name
    assert abs(d_loss.item() - g["d_loss"].item()) < 1e-4
    assert torch.allclose(eng.pred_gt.cpu(), T(g["pred_gt"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(eng.pred_sr.cpu(), T(g["pred_sr"]), rtol=1e-3, atol=1e-4)
    # D's gradients of the discriminator step: fp64-truth criterion.  Truth = the oracle iteration (oracle/steps.py, pinned to the
    # reference's iteration by tests/test_oracle_golden.py) run in fp64 from the same initial states; the fp32 side of the bound is
    # the REFERENCE's stored gradient where the fixture has it in full, the oracle's fp32 run elsewhere.
    from oracle import steps as osteps

    def oracle_iter(dtype):
        tr = osteps.OracleTrainer({k: v.to(dtype) if v.is_floating_point() else v for k, v in g_state0.items()},
                                  {k: v.to(dtype) if v.is_floating_point() else v for k, v in d_state0.items()},
                                  criterions=(("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0)), d_update_interval=1)
        tr.train_step(T(g["gt"]).to(dtype), T(g["lr"]).to(dtype))
        return tr.d_grads()
    d32, d64 = oracle_iter(torch.float32), oracle_iter(torch.float64)
    for n, p in D.named_parameters():
        key = "d_grad/" + n
        assert_fp64_truth(n, p.grad.cpu(), T(g[key]) if key in g.files else d32[n], d64[n])
        if key not in g.files:
            nrm = g[key + "#norm"].item()
            assert abs(nrm - d64[n].norm().item()) <= max(1e-3, 3 * rel_err(d32[n], d64[n])) * d64[n].norm().item(), n
    sd = D.state_dict()
    for k in sd:
        key = "d_state1/" + k
        if key not in g.files:
            continue
        if "num_batches" in k:
            assert int(sd[k]) == int(g[key]) == 3          # D(sr) in the G step + D(gt) + D(sr) in the D step
        else:
            assert torch.allclose(sd[k].cpu(), T(g[key]), rtol=1e-3, atol=2e-4), k
    sg = G.state_dict()
    for k in sg:
        if "num_batches" not in k:
            assert torch.allclose(sg[k].cpu(), T(g["g_state1/" + k]), rtol=1e-3, atol=2e-4), k


def test_train_engine_graph_equals_eager():
    """hipGraph replay of the train step == eager step, bit for bit (same kernels, same order, same optimizer
    code path: both runs use the device-side (capturable) fused Adam - the host-side variant rounds its bias
    corrections differently (1e-7 relative), which GAN dynamics amplify within a few steps)."""
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator

    def run(use_graph):
        cfg = make_cfg(16, 2, 8)
        torch.manual_seed(1)
        D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
        cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
        cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
        cfg.SOLVER.D_UPDATE_INTERVAL = 1
        eng = TrainEngine(cfg, G, D, use_graph=use_graph, adam_capturable=True)
        gen = torch.Generator().manual_seed(2)
        for _ in range(5):
            gt = torch.rand(4, 3, 96, 96, generator=gen).cuda()
            lr = torch.rand(4, 3, 24, 24, generator=gen).cuda()
            eng.step(gt, lr)
        torch.cuda.synchronize()
        assert eng.graph_active == use_graph          # the graph run really replays captured graphs
        return G.state_dict(), D.state_dict(), {k: v.item() for k, v in eng.loss_values.items()}

    g1, d1, l1 = run(False)
    g2, d2, l2 = run(True)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in d1:
        assert torch.equal(d1[k], d2[k]), k
    assert l1 == l2


def test_discriminator_hr192_vs_oracle():
    """BASELINE configs[4]: 192 px crops.  The reference hard-codes 96 px (model.py:31-34, classifier in-features 512*6*6); the
    build derives them from DATA.GT_IMAGE_SIZE (8C * (HR/16)^2 = 73,728).  Logits, input gradient and every parameter
    gradient against the CPU oracle built for the same image size."""
    from oracle import model as om
    from srganst.model import Discriminator
    from srganst.loss import BCEWithLogitsLoss
    cfg = make_cfg()
    cfg.DATA.GT_IMAGE_SIZE = 192
    torch.manual_seed(17)
    D = Discriminator(cfg)
    assert D.state_dict()["classifier.0.weight"].shape == (1024, 73728)
    gen = torch.Generator().manual_seed(18)
    x = torch.rand(2, 3, 192, 192, generator=gen)
    sd = {k: v.clone() for k, v in D.state_dict().items()}
    target = torch.full([2, 1], 0.9)

    def fl(sdx, x_):
        lg = om.discriminator_forward(sdx, x_, True, {})
        return F.binary_cross_entropy_with_logits(lg, target.to(lg.dtype)), lg
    _, g32, (dx32,), (logit_ref,) = oracle_grads(fl, sd, torch.float32, ((x, True),))
    _, g64, (dx64,), _ = oracle_grads(fl, sd, torch.float64, ((x, True),))
    D.cuda().train()
    xg = x.cuda().requires_grad_(True)
    logit = D(xg)
    BCEWithLogitsLoss()(logit, target.cuda()).backward()
    assert torch.allclose(logit.detach().cpu(), logit_ref, rtol=1e-3, atol=1e-4)
    assert_fp64_truth("dx", xg.grad.cpu(), dx32, dx64)
    for n, p in D.named_parameters():
        assert_fp64_truth(n, p.grad.cpu(), g32[n], g64[n])


@pytest.mark.parametrize("interval", [1, 2])
def test_train_engine_pack_reuse_equals_repacking(interval):
    """The engine lets the discriminator step re-use the weight packs made by the generator step's D(sr) (disc_graph._packs).
    Same run with the re-use switched off (every pass packs, as outside an engine): parameters, BatchNorm buffers and losses
    must be bit-identical, with D updated every step and every second step, under hipGraph replay."""
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator

    def run(managed):
        cfg = make_cfg(16, 2, 8)
        torch.manual_seed(1)
        D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
        cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
        cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
        cfg.SOLVER.D_UPDATE_INTERVAL = interval
        eng = TrainEngine(cfg, G, D, use_graph=True, adam_capturable=True)
        if not managed:
            D.__dict__["_packs_managed"] = False
        gen = torch.Generator().manual_seed(2)
        for _ in range(6):
            eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
        torch.cuda.synchronize()
        return G.state_dict(), D.state_dict(), {k: v.item() for k, v in eng.loss_values.items()}

    g1, d1, l1 = run(True)
    g2, d2, l2 = run(False)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in d1:
        assert torch.equal(d1[k], d2[k]), k
    assert l1 == l2


@pytest.mark.parametrize("fail", ["g", "d"])
def test_train_engine_capture_failure_drops_every_graph(fail):
    """A hipGraph bakes in the tensors alive at capture time (D's graph reads the generator graph's static `sr`), so a mixed
    eager / graph engine would replay on stale buffers.  Force the capture of ONE half to fail (an illegal synchronize while
    capturing): the whole engine must fall back to eager - graph_active False, no graph kept - and the run must stay
    bit-identical to an all-eager run.  Runs in a child process (tests/capture_failure_child.py): a failed capture leaves
    torch's capture bookkeeping of that process in an undefined state, which must not leak into the other tests."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capture_failure_child.py")
    r = subprocess.run([sys.executable, child, fail], capture_output=True, text=True, timeout=600)
    assert "FALLBACK-PARITY-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "RECAPTURE-OK" in r.stdout and r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("interval", [1, 2])
def test_train_engine_schedules_are_bit_identical(interval):
    """The iteration's launch schedules (engine.TrainEngine): sequential [G step][D step] (round 1), the discriminator step's two
    passes on two streams (KERNEL.D_TWO_STREAMS), and the default - whole iteration as ONE graph with the discriminator step beside
    the generator's backward (KERNEL.OVERLAP_GD, _iter_gd).  Same kernels, same arguments, same order per tensor: parameters,
    BatchNorm buffers (running statistics move in the order D(sr) of the G step, D(gt), D(sr)) and losses must be bit-identical,
    eager and under hipGraph replay, with D updated every step and every second step."""
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator

    def run(gd, two, use_graph, reuse=True, early=True, defer=True, pack_early=True):
        cfg = make_cfg(16, 2, 8)
        cfg.KERNEL.OVERLAP_GD, cfg.KERNEL.D_TWO_STREAMS, cfg.KERNEL.REUSE_D_SR, cfg.KERNEL.EARLY_D_GT = gd, two, reuse, early
        cfg.KERNEL.EARLY_D_PACK = pack_early
        cfg.KERNEL.DEFER_D_WGRAD = {True: 8, False: 0}.get(defer, defer)      # layers whose last-pass weight gradients move to the other stream
        torch.manual_seed(1)
        D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
        cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
        cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
        cfg.SOLVER.D_UPDATE_INTERVAL = interval
        eng = TrainEngine(cfg, G, D, use_graph=use_graph, adam_capturable=True)
        gen = torch.Generator().manual_seed(2)
        for _ in range(8):
            eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
        torch.cuda.synchronize()
        assert eng.graph_active == use_graph
        sd = {"G." + k: v.clone() for k, v in G.state_dict().items()}
        sd.update({"D." + k: v.clone() for k, v in D.state_dict().items()})
        sd.update({"loss." + k: v.clone() for k, v in eng.loss_values.items()})
        sd["d_loss"] = eng.d_loss.clone()
        return sd

    ref = run(False, False, False)                      # sequential, eager
    assert int(ref["D.features.3.num_batches_tracked"]) == 8 + 2 * (8 // interval)
    # the merged schedule does not run D(sr.detach()) again (KERNEL.REUSE_D_SR: the generator step's D(sr) pass is re-used and the
    # running statistics replayed) - with and without that, against the sequential schedule that runs all three passes
    # ... and starts D(gt)'s forward with the iteration (KERNEL.EARLY_D_GT, statistics replayed in the reference's order)
    for gd, two, use_graph, reuse, early in ((False, False, True, True, True), (False, True, False, True, True), (False, True, True, True, True),
                                             (True, False, False, True, True), (True, False, True, True, True),
                                             (True, False, False, False, False), (True, False, True, False, False),
                                             (True, False, True, True, False), (True, False, True, False, True)):
        out = run(gd, two, use_graph, reuse, early)
        for k in ref:
            assert torch.equal(ref[k], out[k]), (gd, two, use_graph, reuse, early, k)
    # ... and without moving the last pass's weight gradients to the generator's stream (KERNEL.DEFER_D_WGRAD)
    for use_graph, defer in ((False, False), (True, False), (True, 3)):
        out = run(True, False, use_graph, defer=defer)
        for k in ref:
            assert torch.equal(ref[k], out[k]), ("defer", defer, use_graph, k)
    # ... and with D's weights packed in front of D(sr) on the main stream instead of beside the generator's forward (KERNEL.EARLY_D_PACK)
    out = run(True, False, True, pack_early=False)
    for k in ref:
        assert torch.equal(ref[k], out[k]), ("late pack", k)


def test_capture_refuses_second_level_join():
    """What crashed hipStreamEndCapture in round 2 ("a third concurrent branch") is a dependency edge from a second-level stream into a
    first-level one (tools/capture_probe.py); ops.check_capture_join refuses it with CaptureTopologyError.  In a child process."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capture_topology_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=600)
    assert "REFUSED:" in r.stdout and "ALIVE" in r.stdout and r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])


def test_train_engine_schedules_bit_identical_full_width_discriminator():
    """The schedule test above at D_N_CHANNEL = 8 never dispatches the pipelined conv kernel, the MFMA first-layer kernels or the all-taps
    weight-gradient kernel.  Here the discriminator has its full width (64 channels; B = 4: too small for the batched discriminator
    step, so every schedule runs the passes one by one): sequential eager against the merged iteration under hipGraph replay, with
    and without the shared D(sr) pass - parameters, buffers and losses bit for bit."""
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator

    def run(gd, use_graph, reuse):
        cfg = make_cfg(16, 2, 64)
        cfg.KERNEL.OVERLAP_GD, cfg.KERNEL.REUSE_D_SR = gd, reuse
        torch.manual_seed(1)
        D, G = Discriminator(cfg).cuda().train(), Generator(cfg).cuda().train()
        cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
        cfg.add_g_criterion("ST", StructureTensorLoss(), 1 / 3)
        cfg.SOLVER.D_UPDATE_INTERVAL = 1
        eng = TrainEngine(cfg, G, D, use_graph=use_graph, adam_capturable=True)
        gen = torch.Generator().manual_seed(2)
        for _ in range(4):
            eng.step(torch.rand(4, 3, 96, 96, generator=gen).cuda(), torch.rand(4, 3, 24, 24, generator=gen).cuda())
        torch.cuda.synchronize()
        assert eng.graph_active == use_graph and not eng.d_batched
        sd = {"G." + k: v.clone() for k, v in G.state_dict().items()}
        sd.update({"D." + k: v.clone() for k, v in D.state_dict().items()})
        sd.update({"loss." + k: v.clone() for k, v in eng.loss_values.items()})
        sd["d_loss"] = eng.d_loss.clone()
        eng.close()
        return sd
    ref = run(False, False, False)
    for gd, use_graph, reuse in ((True, True, True), (True, True, False)):
        out = run(gd, use_graph, reuse)
        for k in ref:
            assert torch.equal(ref[k], out[k]), (gd, use_graph, reuse, k)

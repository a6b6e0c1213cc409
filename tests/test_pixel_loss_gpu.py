"""GPU parity of the pixel criterions (reference config.py:88-90: nn.MSELoss; BASELINE configs[0] names the pixel-L1 variant,
a one-line config change in the reference): forward value and gradient against torch fp64, stand-alone and inside the fused
criterion sum, plus one BASELINE configs[0] step (SRResNet warm-up, B = 4, pixel-L1 only; warmup.py:74-96) through the engine
against the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_fp64_truth, oracle_grads, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["l1", "mse"])
@pytest.mark.parametrize("shape", [(16, 3, 96, 96), (3, 3, 17, 23), (1, 1, 5, 7)])
def test_pixel_criterion_vs_fp64(kind, shape):
    from srganst.loss import L1Loss, MSELoss
    crit = L1Loss() if kind == "l1" else MSELoss()
    g = torch.Generator().manual_seed(sum(shape))
    gt = torch.rand(*shape, generator=g)
    x = (gt + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    x[..., 0, 0] = gt[..., 0, 0]                           # exact ties: sign(0) = 0 in torch's L1 gradient
    x64 = x.double().requires_grad_(True)
    ref = (F.l1_loss if kind == "l1" else F.mse_loss)(x64, gt.double())
    (gref,) = torch.autograd.grad(ref * 0.7, x64)
    xg = x.cuda().requires_grad_(True)
    loss = crit(xg, gt.cuda())
    (gx,) = torch.autograd.grad(loss * 0.7, xg)           # non-unit upstream gradient
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_err(gx.cpu(), gref) < 1e-6
    if kind == "l1":
        assert torch.equal(gx.cpu() == 0, gref == 0)      # ties give exactly zero


def test_l1_inside_fused_criterion_sum():
    """L1 + structure tensor / 3 as ONE autograd node (loss.criterion_sum) = the sum of the stand-alone criterions."""
    from srganst.loss import L1Loss, StructureTensorLoss, criterion_sum
    g = torch.Generator().manual_seed(4)
    gt = torch.rand(4, 3, 96, 96, generator=g).cuda()
    x = (gt + 0.05 * torch.randn(4, 3, 96, 96, generator=g).cuda()).clamp(0, 1)
    l1, st = L1Loss(), StructureTensorLoss()
    xa = x.clone().requires_grad_(True)
    total, weighted = criterion_sum(xa, gt, [l1, st], [1.0, 1.0 / 3.0])
    (ga,) = torch.autograd.grad(total, xa)
    xb = x.clone().requires_grad_(True)
    ref = l1(xb, gt) + st(xb, gt) * (1.0 / 3.0)
    (gb,) = torch.autograd.grad(ref, xb)
    assert abs(total.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert abs(weighted[0].item() - l1(x, gt).item()) <= 1e-6
    assert rel_err(ga.cpu(), gb.cpu()) < 1e-6


@pytest.mark.parametrize("kind", ["mse", "l1"])
@pytest.mark.parametrize("shape", [(16, 3, 96, 96), (2, 3, 41, 75), (8, 3, 192, 192)])
def test_pixel_criterion_riding_in_the_structure_tensor_kernels(kind, shape, monkeypatch):
    """Pixel + structure-tensor pair of the training step (train.py:129-140) as two launches in all (sst_st_pixel_loss_fwd / _bwd:
    the pixel criterion rides along in the structure-tensor kernels) against the four-launch form (SST_FUSE_PIXEL_ST=0) and against
    torch fp64 for the pixel part; either order of the terms, non-unit upstream gradient, ragged image sizes."""
    from srganst import loss as sloss
    from srganst.loss import L1Loss, MSELoss, StructureTensorLoss, criterion_sum
    g = torch.Generator().manual_seed(sum(shape) + len(kind))
    gt = torch.rand(*shape, generator=g)
    x = (gt + 0.05 * torch.randn(*shape, generator=g)).clamp(0, 1)
    pix, st = (MSELoss() if kind == "mse" else L1Loss()), StructureTensorLoss()
    xd, gtd = x.cuda(), gt.cuda()
    for terms, w in (([pix, st], [1.0, 1.0 / 3.0]), ([st, pix], [0.25, 2.0])):
        outs = []
        for fuse in (True, False):
            monkeypatch.setattr(sloss, "FUSE_PIXEL_INTO_ST", fuse)
            xa = xd.clone().requires_grad_(True)
            total, weighted = criterion_sum(xa, gtd, terms, w)
            (ga,) = torch.autograd.grad(total * 0.7, xa)
            outs.append((total.item(), weighted.cpu(), ga.cpu()))
        (t1, w1, g1), (t0, w0, g0) = outs
        assert abs(t1 - t0) <= 2e-6 * abs(t0) and torch.allclose(w1, w0, rtol=2e-6, atol=0)
        assert rel_err(g1, g0) < 1e-6
        pi = terms.index(pix)
        ref = (F.mse_loss if kind == "mse" else F.l1_loss)(x.double(), gt.double()) * w[pi]
        assert abs(w1[pi].item() - ref.item()) <= 1e-5 * abs(ref.item())


def test_configs0_warmup_step_l1_vs_oracle():
    """BASELINE configs[0] on the HIP path: one warm-up iteration with the pixel-L1 criterion only, B = 4, 96 -> 24 px
    (reduced depth to keep the CPU oracle short): SR, loss, every gradient (fp64-truth rule) and the Adam-updated weights."""
    from oracle import model as om
    from oracle import steps as osteps
    from srganst.config import Config
    from srganst.engine import WarmupEngine
    from srganst.loss import L1Loss
    from srganst.model import Generator
    cfg = Config()
    cfg.MODEL.G_N_RCB = 3
    torch.manual_seed(21)
    G = Generator(cfg)
    sd0 = {k: v.clone() for k, v in G.state_dict().items()}
    gen = torch.Generator().manual_seed(22)
    gt = torch.rand(4, 3, 96, 96, generator=gen)
    lr = torch.rand(4, 3, 24, 24, generator=gen)

    def fl(sdx, lr_, gt_):
        sr_ = om.generator_forward(sdx, lr_, True, {})
        return F.l1_loss(sr_, gt_), sr_
    ins = ((lr, False), (gt, False))
    l32, g32, _, (sr_ref,) = oracle_grads(fl, sd0, torch.float32, ins)
    _, g64, _, _ = oracle_grads(fl, sd0, torch.float64, ins)
    tr = osteps.OracleTrainer(sd0, criterions=(("Pixel", 1.0),), pixel_kind="l1")
    tr.warmup_step(gt, lr)
    G.cuda().train()
    eng = WarmupEngine(cfg, G, {"Pixel": L1Loss()}, {"Pixel": 1.0}, use_graph=False)
    vals = eng.step(gt.cuda(), lr.cuda())
    assert rel_err(eng.sr.cpu(), sr_ref) < 1e-3
    assert abs(vals["Pixel"].item() - l32.item()) <= 1e-3 * abs(l32.item())
    for n, p in G.named_parameters():
        assert_fp64_truth(n, p.grad.cpu(), g32[n], g64[n])
    sd1 = G.state_dict()
    for k in om.param_keys(sd0):
        # Adam with eps 1e-4 turns gradient noise of tiny gradients into sign-sized steps; compare the step taken
        assert torch.allclose(sd1[k].cpu() - sd0[k], tr.g[k].detach() - sd0[k], rtol=1e-2, atol=2e-5), k

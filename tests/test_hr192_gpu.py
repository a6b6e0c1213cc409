"""BASELINE configs[4] on the HIP path: one complete SRGAN iteration (reference train.py:116-164: generator update through the
frozen discriminator, then the discriminator update on gt and sr.detach()) at 192-px HR crops, B = 2, reduced generator depth.
The reference hard-codes 96 px in the discriminator (model.py:31-34, 61: classifier in-features 512*6*6); the build derives the
classifier from DATA.GT_IMAGE_SIZE (512 * 12 * 12 = 73,728 inputs).  Checked against oracle/steps.py run in fp32 and fp64:
SR, every loss term, both logit sets and every gradient of both networks (fp64-truth rule of conftest.py)."""
import pytest
import torch

from conftest import assert_fp64_truth, rel_err

pytestmark = pytest.mark.gpu


def test_train_iteration_hr192_vs_oracle():
    from oracle import model as om
    from oracle import steps as osteps
    from srganst.config import Config
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    cfg = Config()
    cfg.DATA.GT_IMAGE_SIZE = 192
    cfg.MODEL.G_N_RCB = 2
    torch.manual_seed(31)
    D = Discriminator(cfg)
    G = Generator(cfg)
    assert D.state_dict()["classifier.0.weight"].shape == (1024, 73728)
    g0 = {k: v.clone() for k, v in G.state_dict().items()}
    d0 = {k: v.clone() for k, v in D.state_dict().items()}
    gen = torch.Generator().manual_seed(32)
    gt = torch.rand(2, 3, 192, 192, generator=gen)
    lr = torch.rand(2, 3, 48, 48, generator=gen)
    crits = (("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0))

    def oracle_iter(dtype):
        cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        tr = osteps.OracleTrainer(cast(g0), cast(d0), criterions=crits, d_update_interval=1)
        sr, losses, d_loss = tr.train_step(gt.to(dtype), lr.to(dtype))
        return sr, losses, d_loss, tr.g_grads(), tr.d_grads()
    sr32, l32, dl32, gg32, dg32 = oracle_iter(torch.float32)
    _, _, _, gg64, dg64 = oracle_iter(torch.float64)

    D.cuda().train()
    G.cuda().train()
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1.0 / 3.0)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=False)
    losses, d_loss = eng.step(gt.cuda(), lr.cuda())
    assert eng.sr.shape == (2, 3, 192, 192) and rel_err(eng.sr.cpu(), sr32) < 1e-3
    for name in ("Adversarial", "Pixel", "ST"):
        assert abs(losses[name].item() - l32[name].item()) <= 1e-3 * abs(l32[name].item()), name
    assert abs(d_loss.item() - dl32.item()) <= 1e-3 * abs(dl32.item())
    for n, p in G.named_parameters():
        assert_fp64_truth("G." + n, p.grad.cpu(), gg32[n], gg64[n])
    for n, p in D.named_parameters():
        assert_fp64_truth("D." + n, p.grad.cpu(), dg32[n], dg64[n])
    assert int(D.state_dict()["features.3.num_batches_tracked"]) == 3      # D(sr) in the G step + D(gt) + D(sr)

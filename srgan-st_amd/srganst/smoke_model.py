"""One tiny generator fwd+bwd step on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import torch


def run():
    from oracle import model as om
    from .config import Config
    from .loss import MSELoss, StructureTensorLoss
    from .model import Generator
    cfg = Config()
    cfg.MODEL.G_N_CHANNEL, cfg.MODEL.G_N_RCB = 64, 2          # full width: the trunk runs the band kernels in accumulator mode
    torch.manual_seed(0)
    G = Generator(cfg)
    gen = torch.Generator().manual_seed(1)
    gt = torch.rand(2, 3, 96, 96, generator=gen)
    lr = torch.rand(2, 3, 24, 24, generator=gen)
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    for k in om.param_keys(sd):
        sd[k].requires_grad_(True)
    from oracle import st as ost
    sr_ref = om.generator_forward(sd, lr, True, {})
    (torch.nn.functional.mse_loss(sr_ref, gt) + ost.st_loss(sr_ref, gt) / 3).backward()
    from . import _abi
    n0 = (_abi.lib().sst_debug_band_launches(), _abi.lib().sst_debug_wgrad_band_launches())
    G.to("cuda:0").train()
    sr = G(lr.to("cuda:0"))
    loss = MSELoss()(sr, gt.to("cuda:0")) + StructureTensorLoss()(sr, gt.to("cuda:0")) * (1 / 3)
    loss.backward()
    torch.cuda.synchronize()
    e_sr = ((sr.detach().cpu().double() - sr_ref.detach().double()).norm() / sr_ref.detach().double().norm()).item()
    worst = 0.0
    for n, p in G.named_parameters():
        r = sd[n].grad.double()
        worst = max(worst, ((p.grad.cpu().double() - r).norm() / r.norm().clamp_min(1e-30)).item())
    assert e_sr < 1e-3 and worst < 5e-3, (e_sr, worst)
    nb = _abi.lib().sst_debug_band_launches() - n0[0]
    assert nb >= 8, f"the band conv kernel did not run ({nb} launches)"
    print(f"smoke ok: generator step (64 ch, 2 blocks, 96 px) SR rel err {e_sr:.2e}, worst param-grad rel err {worst:.2e}; "
          f"{nb} band-conv launches")

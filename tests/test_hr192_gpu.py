"""BASELINE configs[4] on the HIP path: one complete SRGAN iteration (reference train.py:116-164: generator update through the
frozen discriminator, then the discriminator update on gt and sr.detach()) at 192-px HR crops, B = 2, reduced generator depth.
The reference hard-codes 96 px in the discriminator (model.py:31-34, 61: classifier in-features 512*6*6); the build derives the
classifier from DATA.GT_IMAGE_SIZE (512 * 12 * 12 = 73,728 inputs).  Checked against oracle/steps.py run in fp32 and fp64:
SR, every loss term, both logit sets and every gradient of both networks (fp64-truth rule of conftest.py)."""
import pytest
import torch

from conftest import TRUTH_FACTOR, TRUTH_FLOOR, assert_fp64_truth, rel_err

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


def _clamp_flips(a, b):
    return int(((a == 0) != (b == 0)).sum() + ((a == 1) != (b == 1)).sum())


_BENCH_ORACLE = {}


@pytest.mark.parametrize("schedule", ["shared", "batched"])
def test_train_iteration_bench_size_vs_oracle(schedule):
    """schedule "shared": the engine's default (the discriminator step works on the generator step's D(sr) pass, KERNEL.REUSE_D_SR: kept
    in a two-pass arena, D(gt) joins it, one backward over both); "batched": the bench headline's schedule - all three discriminator
    forwards, the discriminator step's two passes as ONE batch of 2B images with per-pass BatchNorm statistics (KERNEL.BATCH_D_STEP).
    In both the weight gradients sum over 32 images in one kernel: the parity statement IS this fp64-truth rule (no bit-identity with
    the pass-by-pass schedule is claimed).

    The bench line's own workload on the bench line's own code path: full-size generator (16 residual blocks) and discriminator,
    B = 16, 96-px crops, the default engine schedule (whole iteration as one launch DAG, discriminator step on a side stream;
    eager here, the graph replays the same launches - test_train_engine_schedules_are_bit_identical).  SR, every loss term, d_loss
    and every gradient of both networks against oracle/steps.py run in fp64.

    What the criterion has to live with at this size (measured, tools/grad_errors_iter.py / tools/tail_errors.py):
      * clamp_(0,1) makes the gradient DISCONTINUOUS in the forward values: one output pixel whose pre-clamp value crosses 0 or 1
        between two fp32 runs (expected count ~ 1 per run at 442k outputs with 1.4e-6 forward error) removes one of ~117k terms
        and moves every trunk gradient by ~1/sqrt(117k) = 3e-3.  The input is therefore the first seed on which the HIP forward
        and the fp64 forward clamp the same pixels (forward passes only; the seed and the flip counts are reported);
      * torch CPU accumulates BatchNorm-backward sums and elementwise BatchNorm arithmetic in double (acc_type), which no fp32 GPU
        path does: the same plain-torch graph run on the GPU (MIOpen / rocBLAS fp32) is 1-3x less accurate than the CPU oracle and
        fails the CPU-only 3x rule on 6-40 of 153 parameters depending on the seed.  The yardstick per parameter is therefore the
        LARGER error of the two fp32 references (oracle on CPU, oracle on the GPU), both against fp64.
    Criterion: whole-network gradient (all parameters as one vector) within max(1e-3, 3 x yardstick) for G and for D; per parameter
    the same bound for at least 95 % of the parameters and 3 x the bound for every one of them (a wrong kernel is off by O(1);
    measured in rounds 2 and 3: no parameter beyond the 1 x bound on either schedule)."""
    from oracle import model as om
    from oracle import steps as osteps
    from srganst.config import Config
    from srganst.engine import TrainEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    cfg = Config()
    assert cfg.KERNEL.OVERLAP_GD and cfg.KERNEL.BATCH_D_STEP
    cfg.KERNEL.REUSE_D_SR = schedule == "shared"
    torch.manual_seed(41)
    D = Discriminator(cfg)
    G = Generator(cfg)
    g0 = {k: v.clone() for k, v in G.state_dict().items()}
    d0 = {k: v.clone() for k, v in D.state_dict().items()}
    D.cuda().train()
    G.cuda().train()
    crits = (("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0))
    g64sd = {k: (v.double() if v.is_floating_point() else v) for k, v in g0.items()}

    chosen = None
    for seed in range(42, 58):
        gen = torch.Generator().manual_seed(seed)
        gt = torch.rand(16, 3, 96, 96, generator=gen)
        lr = torch.rand(16, 3, 24, 24, generator=gen)
        with torch.no_grad():
            sr_h = G(lr.cuda()).cpu()
            sr_64 = om.generator_forward(g64sd, lr.double(), True, {})
        G.load_state_dict(g0)                                     # the probe moved the BatchNorm buffers
        flips = _clamp_flips(sr_h, sr_64)
        print(f"seed {seed}: {flips} clamp flips between the HIP and the fp64 forward")
        if flips == 0:
            chosen = seed
            break
    assert chosen is not None, "no input in 16 seeds on which the HIP and the fp64 forward clamp the same pixels"

    def oracle_iter(dtype, device="cpu"):
        cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v).to(device) for k, v in sd.items()}
        tr = osteps.OracleTrainer(cast(g0), cast(d0), criterions=crits, d_update_interval=1)
        sr, losses, d_loss = tr.train_step(gt.to(dtype).to(device), lr.to(dtype).to(device))
        cpu = lambda d: {k: v.cpu() for k, v in d.items()}
        return sr.cpu(), {k: v.cpu() for k, v in losses.items()}, d_loss.cpu(), cpu(tr.g_grads()), cpu(tr.d_grads())
    if chosen not in _BENCH_ORACLE:                               # the oracle runs are the same for both schedules
        _BENCH_ORACLE[chosen] = (oracle_iter(torch.float32), oracle_iter(torch.float64),
                                 oracle_iter(torch.float32, "cuda"))      # the same plain-torch graph on the GPU: second fp32 reference
    (sr32, l32, dl32, gg32, dg32), (sr64, _, _, gg64, dg64), (srm, _, _, ggm, dgm) = _BENCH_ORACLE[chosen]

    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1.0 / 3.0)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    eng = TrainEngine(cfg, G, D, use_graph=False)
    losses, d_loss = eng.step(gt.cuda(), lr.cuda())
    assert eng.d_batched and eng.d_sr_reused == (schedule == "shared")      # one backward over 2B images in both schedules
    assert rel_err(eng.sr.cpu(), sr32) < 1e-3
    print(f"clamp flips against fp64 on seed {chosen}: hip {_clamp_flips(eng.sr.cpu(), sr64)}, oracle fp32 {_clamp_flips(sr32, sr64)}, "
          f"oracle on the GPU {_clamp_flips(srm, sr64)}")
    for name in ("Adversarial", "Pixel", "ST"):
        assert abs(losses[name].item() - l32[name].item()) <= 1e-3 * abs(l32[name].item()), name
    assert abs(d_loss.item() - dl32.item()) <= 1e-3 * abs(dl32.item())

    for tag, module, r32, rgm, r64 in (("G", G, gg32, ggm, gg64), ("D", D, dg32, dgm, dg64)):
        names = [n for n, _ in module.named_parameters()]
        hip = {n: p.grad.cpu() for n, p in module.named_parameters()}
        flat = lambda d: torch.cat([d[n].double().flatten() for n in names])
        yard = max(rel_err(flat(r32), flat(r64)), rel_err(flat(rgm), flat(r64)))
        e = rel_err(flat(hip), flat(r64))
        print(f"{tag}: whole-network gradient |hip - fp64| = {e:.3e}, oracle fp32 {rel_err(flat(r32), flat(r64)):.3e}, "
              f"oracle on the GPU {rel_err(flat(rgm), flat(r64)):.3e}")
        assert e <= max(TRUTH_FLOOR, TRUTH_FACTOR * yard), f"{tag}: whole-network gradient {e:.3e} > max(1e-3, 3 x {yard:.3e})"
        beyond = []
        for n in names:
            e = rel_err(hip[n], r64[n])
            bound = max(TRUTH_FLOOR, TRUTH_FACTOR * max(rel_err(r32[n], r64[n]), rel_err(rgm[n], r64[n])))
            assert e <= 3 * bound, f"{tag}.{n}: |hip - fp64| = {e:.3e} > 3 x max(1e-3, 3 x fp32 references) = {3 * bound:.3e}"
            if e > bound:
                beyond.append((n, e, bound))
        print(f"{tag}: {len(beyond)} of {len(names)} parameters beyond max(1e-3, 3 x fp32 references): {beyond[:6]}")
        assert len(beyond) <= 0.05 * len(names), f"{tag}: {len(beyond)} of {len(names)} parameters beyond the bound: {beyond[:8]}"
    eng.close()

"""GPU parity of the whole generator path (HIP kernels through the C ABI) against the golden
vectors captured from the reference: SR tensor, each loss, every parameter gradient, Adam-updated
parameters and BN buffers (reduced config), and the seed-pinned full-size model."""
import numpy as np
import pytest
import torch

from conftest import assert_fp64_truth, oracle_grads, rel_err

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def make_cfg(ch=64, rcb=16, dch=64):
    from srganst.config import Config
    cfg = Config()
    cfg.MODEL.G_N_CHANNEL = ch
    cfg.MODEL.G_N_RCB = rcb
    cfg.MODEL.D_N_CHANNEL = dch
    return cfg


def test_generator_small_two_adam_steps(golden):
    from srganst.model import Generator
    from srganst.loss import MSELoss, StructureTensorLoss
    g = golden("g_small_step")
    G = Generator(make_cfg(8, 2))
    sd0 = {k[len("state0/"):]: T(g[k]) for k in g.files if k.startswith("state0/")}
    assert set(sd0) == set(G.state_dict().keys())          # state-dict key set == reference's
    G.load_state_dict(sd0)
    G.cuda().train()
    opt = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-4)
    mse, st = MSELoss(), StructureTensorLoss()
    gt, lr = T(g["gt"]).cuda(), T(g["lr"]).cuda()
    for step in range(2):
        G.zero_grad()
        sr = G(lr)
        l_pix = mse(sr, gt) * 1.0
        l_st = st(sr, gt) * (1 / 3)
        (l_pix + l_st).backward()
        if step == 0:
            # north-star tolerance: 1e-3 relative fp32
            assert rel_err(sr.detach().cpu(), g["sr0"]) < 1e-4
            assert abs(l_pix.item() - g["loss_pixel0"].item()) < 1e-4 * abs(g["loss_pixel0"].item())
            assert abs(l_st.item() - g["loss_st0"].item()) < 1e-3 * abs(g["loss_st0"].item())
            # fp64-truth criterion; fp32 side of the bound = the reference's own stored gradient
            from oracle import model as om
            from oracle import st as ost

            def fl(sdx, lr_, gt_):
                sr_ = om.generator_forward(sdx, lr_, True, {})
                return torch.nn.functional.mse_loss(sr_, gt_) + ost.st_loss(sr_, gt_) / 3
            _, g64, _, _ = oracle_grads(fl, sd0, torch.float64, ((T(g["lr"]), False), (T(g["gt"]), False)))
            for n, p in G.named_parameters():
                assert_fp64_truth(n, p.grad.cpu(), T(g["grad0/" + n]), g64[n])
        opt.step()
        sd = G.state_dict()
        for k in sd:
            ref = T(g[f"state{step+1}/" + k])
            if "num_batches" in k:
                assert int(sd[k]) == int(ref)
            else:
                assert torch.allclose(sd[k].cpu(), ref, rtol=1e-3, atol=1e-4), (step, k)


def test_generator_full_seed0(golden):
    from srganst.model import Generator
    from srganst.loss import MSELoss
    g = golden("g_full_seed0")
    torch.manual_seed(0)
    G = Generator(make_cfg())
    assert sum(p.numel() for p in G.parameters()) == 1547350            # reference model.py:193
    sd = G.state_dict()
    for k in [f[2:] for f in g.files if f.startswith("w/")]:
        assert torch.equal(sd[k], T(g["w/" + k])), k                    # same RNG order as the reference
    sd0 = {k: v.clone() for k, v in sd.items()}
    G.cuda().train()
    sr = G(T(g["lr"]).cuda())
    assert sr.shape == (2, 3, 96, 96) and float(sr.min()) >= 0 and float(sr.max()) <= 1
    assert rel_err(sr.detach().cpu(), g["sr"]) < 1e-3
    loss = MSELoss()(sr, T(g["gt"]).cuda())
    assert abs(loss.item() - g["loss"].item()) < 1e-3 * g["loss"].item()
    loss.backward()
    # gradients: fp64-truth criterion (conftest.assert_fp64_truth).  The truth is the oracle (pinned to the reference by
    # tests/test_oracle_golden.py) run in fp64 on the same weights / inputs; the fp32 side of the bound is the REFERENCE's own
    # fp32 gradient where the fixture stores it (g/...), the oracle's fp32 run elsewhere.
    from oracle import model as om

    def fl(sdx, lr, gt):
        return torch.nn.functional.mse_loss(om.generator_forward(sdx, lr, True, {}), gt)
    ins = ((T(g["lr"]), False), (T(g["gt"]), False))
    _, g32, _, _ = oracle_grads(fl, sd0, torch.float32, ins)
    _, g64, _, _ = oracle_grads(fl, sd0, torch.float64, ins)
    named = dict(G.named_parameters())
    stored = {f[2:] for f in g.files if f.startswith("g/")}
    report = []
    for k, v in named.items():
        ref32 = T(g["g/" + k]) if k in stored else g32[k]
        assert_fp64_truth(k, v.grad.cpu(), ref32, g64[k], report)
    worst = max(report, key=lambda r: r[1])
    print(f"worst param-grad error vs fp64: {worst[0]} {worst[1]:.2e} (oracle fp32: {worst[2]:.2e})")
    sd = G.state_dict()
    assert torch.allclose(sd["trunk.0.rcb.1.running_mean"].cpu(), T(g["bn/trunk.0.rcb.1.running_mean"]), rtol=1e-3, atol=1e-6)
    assert torch.allclose(sd["trunk.0.rcb.1.running_var"].cpu(), T(g["bn/trunk.0.rcb.1.running_var"]), rtol=1e-3, atol=1e-6)


def test_generator_vs_oracle_fresh_input():
    """Fresh seeded batch at the bench shape (B=16, 24->96): SR and grads vs the CPU oracle."""
    from oracle import model as om
    from srganst.model import Generator
    from srganst.loss import MSELoss
    torch.manual_seed(3)
    G = Generator(make_cfg(64, 4))
    gen = torch.Generator().manual_seed(11)
    lr = torch.rand(16, 3, 24, 24, generator=gen)
    gt = torch.rand(16, 3, 96, 96, generator=gen)
    sd = {k: v.clone() for k, v in G.state_dict().items()}

    def fl(sdx, lr_, gt_):
        sr_ = om.generator_forward(sdx, lr_, True, {})
        return torch.nn.functional.mse_loss(sr_, gt_), sr_
    ins = ((lr, False), (gt, False))
    _, g32, _, (sr_ref,) = oracle_grads(fl, sd, torch.float32, ins)
    _, g64, _, _ = oracle_grads(fl, sd, torch.float64, ins)
    G.cuda().train()
    sr = G(lr.cuda())
    MSELoss()(sr, gt.cuda()).backward()
    assert rel_err(sr.detach().cpu(), sr_ref) < 1e-3
    for n, p in G.named_parameters():
        assert_fp64_truth(n, p.grad.cpu(), g32[n], g64[n])


def test_generator_eval_and_no_cpu_fallback():
    from srganst import _abi
    from srganst.model import Generator
    from oracle import model as om
    torch.manual_seed(5)
    G = Generator(make_cfg(16, 2))
    with pytest.raises(_abi.HipPathError):
        G(torch.rand(1, 3, 8, 8))                       # CPU tensor: loud failure, no fallback
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    x = torch.rand(1, 3, 17, 23)                        # arbitrary (validation-style) size, batch 1
    ref = om.generator_forward(sd, x, training=False)
    G.cuda().eval()
    with torch.no_grad():
        out = G(x.cuda())
    assert out.shape == (1, 3, 68, 92)
    assert rel_err(out.cpu(), ref) < 1e-4


def test_accumulator_mode_equals_partial_tile_mode(monkeypatch):
    """BatchNorm statistics through fp64 atomic accumulators + consumer-side finalize (csrc/conv_epilogue.h: BandAcc) against
    the partial-tile + bn_finalize path: same SR, gradients and BatchNorm buffers up to fp32 rounding, over two steps."""
    from srganst.model import Generator
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst import ops
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(9)
    gt = torch.rand(4, 3, 96, 96, generator=g).cuda()
    lr = torch.rand(4, 3, 24, 24, generator=g).cuda()
    res = {}
    G0 = Generator(make_cfg(64, 3)).cuda()
    sd0 = {k: v.clone() for k, v in G0.state_dict().items()}
    assert ops.conv_acc_supported(4, 24, 24, 64, 64)
    for mode in ("1", "0"):
        monkeypatch.setenv("SST_ATOMIC_STATS", mode)
        G = Generator(make_cfg(64, 3)).cuda()
        G.load_state_dict(sd0)
        G.train()
        mse, st = MSELoss(), StructureTensorLoss()
        out = []
        for step in range(2):
            G.zero_grad()
            sr = G(lr)
            (mse(sr, gt) + st(sr, gt) * (1 / 3)).backward()
            out.append((sr.detach().cpu(), {n: p.grad.cpu().clone() for n, p in G.named_parameters()},
                        {k: v.cpu().clone() for k, v in G.state_dict().items() if "running" in k}))
        res[mode] = out
    for (sr_a, gr_a, bn_a), (sr_b, gr_b, bn_b) in zip(res["1"], res["0"]):
        assert rel_err(sr_a, sr_b) < 1e-5
        for n in gr_a:
            tol = 1e-3      # two fp32 orderings of the statistics seen through 3 blocks' backward: 2.4e-4 .. 5.0e-4 on conv1.0.weight
            # (the longest path) depending on the finalize kernel's summation order; scalar slope gradients are cancelling sums
            assert rel_err(gr_a[n], gr_b[n]) < tol, n
        for k in bn_a:
            assert torch.allclose(bn_a[k], bn_b[k], rtol=1e-5, atol=1e-7), k


def test_generator_vs_oracle_hr192():
    """BASELINE configs[4] shape (48 -> 192 px HR crops): the trunk runs the band kernel with one-row bands (W = 48) and the
    up-sampler the 64x64-tile kernel at 96 px; SR, both losses and every gradient against the CPU oracle."""
    from oracle import model as om
    from oracle import st as ost
    from srganst import _abi
    from srganst.model import Generator
    from srganst.loss import MSELoss, StructureTensorLoss
    torch.manual_seed(13)
    G = Generator(make_cfg(64, 2))
    gen = torch.Generator().manual_seed(14)
    lr = torch.rand(2, 3, 48, 48, generator=gen)
    gt = torch.rand(2, 3, 192, 192, generator=gen)
    sd = {k: v.clone() for k, v in G.state_dict().items()}

    def fl(sdx, lr_, gt_):
        sr_ = om.generator_forward(sdx, lr_, True, {})
        return torch.nn.functional.mse_loss(sr_, gt_) + ost.st_loss(sr_, gt_) / 3, sr_
    ins = ((lr, False), (gt, False))
    l_ref, g32, _, (sr_ref,) = oracle_grads(fl, sd, torch.float32, ins)
    _, g64, _, _ = oracle_grads(fl, sd, torch.float64, ins)
    n0 = _abi.lib().sst_debug_band_launches()
    G.cuda().train()
    sr = G(lr.cuda())
    loss = MSELoss()(sr, gt.cuda()) + StructureTensorLoss()(sr, gt.cuda()) * (1 / 3)
    loss.backward()
    assert _abi.lib().sst_debug_band_launches() - n0 >= 8          # 5 forward + 5 backward trunk convs
    assert sr.shape == (2, 3, 192, 192)
    assert rel_err(sr.detach().cpu(), sr_ref.detach()) < 1e-3
    assert abs(loss.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    for n, p in G.named_parameters():
        assert_fp64_truth(n, p.grad.cpu(), g32[n], g64[n])


def test_accumulator_mode_run_to_run(capsys):
    """What is promised about reproducibility at the full width (64 channels: the trunk runs conv_band_kernel in accumulator
    mode, BatchNorm sums through hardware fp64 atomics, csrc/conv_epilogue.h: BandAcc).  The atomic adds commute but do not
    associate: two runs of the same step may differ in the last bits of an fp64 sum (1e-16 relative), which can - rarely - move
    an fp32 rounding of a BatchNorm mean / rstd.  ASSERTED: run-to-run differences of SR, every gradient and every BatchNorm buffer
    stay below 1e-6 relative (three orders under the 1e-3 parity contract).  NOT asserted: bitwise equality (reported: it is
    what is observed in practice; the partial-tile path, SST_ATOMIC_STATS=0, and every kernel outside the trunk are bitwise
    reproducible by construction - fixed-order reductions, tests/test_discriminator_gpu.py::test_train_engine_graph_equals_eager)."""
    from srganst.model import Generator
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst import ops
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(19)
    gt = torch.rand(16, 3, 96, 96, generator=g).cuda()
    lr = torch.rand(16, 3, 24, 24, generator=g).cuda()
    assert ops.conv_acc_supported(16, 24, 24, 64, 64)
    G0 = Generator(make_cfg(64, 4)).cuda()
    sd0 = {k: v.clone() for k, v in G0.state_dict().items()}
    runs = []
    for _ in range(3):
        G = Generator(make_cfg(64, 4)).cuda()
        G.load_state_dict(sd0)
        G.train()
        mse, st = MSELoss(), StructureTensorLoss()
        sr = G(lr)
        (mse(sr, gt) + st(sr, gt) * (1 / 3)).backward()
        out = {"sr": sr.detach().clone()}
        out.update({"grad/" + n: p.grad.clone() for n, p in G.named_parameters()})
        out.update({"buf/" + k: v.clone() for k, v in G.state_dict().items() if "running" in k})
        runs.append(out)
    worst, identical = 0.0, 0
    for k in runs[0]:
        for other in runs[1:]:
            a, b = runs[0][k].double(), other[k].double()
            d = float((a - b).norm() / a.norm().clamp_min(1e-30))
            worst = max(worst, d)
            identical += int(torch.equal(runs[0][k], other[k]))
    with capsys.disabled():
        print(f"\\n[accumulator mode, 3 runs] worst run-to-run difference {worst:.2e}; {identical} of {2 * len(runs[0])} tensor pairs bit-identical")
    assert worst < 1e-6

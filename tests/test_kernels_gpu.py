"""GPU parity of the individual HIP kernels (through the C ABI) against plain torch fp32/fp64 on CPU.

Tolerance: the north star allows 1e-3 relative fp32; fp32 MFMA is an exact fp32 FMA chain so we
hold the kernels to 2e-5 norm-wise against an fp64 reference (re-association only)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def ops():
    from srganst import ops
    return ops


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride
    (2, 24, 24, 64, 64, 3, 1),
    (1, 13, 9, 8, 8, 3, 1),        # ragged tile edges, tiny channels (reduced golden config)
    (2, 12, 12, 64, 256, 3, 1),
    (2, 24, 24, 64, 64, 3, 2),
    (1, 12, 12, 128, 256, 3, 2),   # two input-channel blocks
    (1, 11, 7, 3, 64, 3, 1),       # Cin not a multiple of 4
    (2, 24, 24, 3, 64, 9, 1),      # conv1
    (1, 32, 24, 64, 3, 9, 1),      # conv3
    (1, 16, 16, 256, 64, 3, 1),    # up-conv dgrad shape
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_plain(ops, case):
    B, H, W, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), s, k // 2)
    wp = ops.pack_conv(w.cuda(), 0)
    y, _, _, _ = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, k, s, bias=b.cuda())
    assert rel_err(nchw(y.cpu()), ref) < TOL


def test_conv_fwd_fused_prologue_stats_residual(ops):
    B, H, W, C = 2, 24, 24, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / 24.0
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    slope = torch.tensor([0.25])
    res = torch.randn(B, C, H, W, generator=g)
    xin = F.prelu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), slope.double())
    conv = F.conv2d(xin, w.double(), None, 1, 1)
    ref = conv + res.double()
    y, _, stats, cnt = ops.conv_fwd(nhwc(x).cuda(), ops.pack_conv(w.cuda()), C, 3, 1, in_scale=sc.cuda(), in_shift=sh.cuda(),
                                    in_slope=slope.cuda(), in_act=ops.ACT_SLOPE, residual=nhwc(res).cuda(), want_stats=True)
    assert rel_err(nchw(y.cpu()), ref) < TOL
    # BN finalize on the stats of y (stats are taken on the stored value, i.e. incl. the residual)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    mean, rstd, scale, shift = ops.bn_finalize(stats, cnt, gamma.cuda(), beta.cuda(), (rm_d := rm.cuda()), (rv_d := rv.cuda()))
    m_ref = ref.mean(dim=(0, 2, 3))
    v_ref = ref.var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(mean.cpu(), m_ref) < TOL
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(v_ref + 1e-5)) < TOL
    assert rel_err(scale.cpu(), gamma.double() / torch.sqrt(v_ref + 1e-5)) < TOL
    assert rel_err(shift.cpu(), beta.double() - m_ref * gamma.double() / torch.sqrt(v_ref + 1e-5)) < 1e-4
    bn = torch.nn.BatchNorm2d(C)
    bn.train()
    bn(ref.float())
    assert torch.allclose(rm_d.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv_d.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)


def test_bn_stats_large_mean(ops):
    """Chan-combined centred statistics stay accurate when |mean| >> std (naive E[x^2]-E[x]^2 would not)."""
    B, H, W, C = 4, 24, 24, 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.zeros(C, C, 3, 3)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    bias = torch.full((C,), 100.0)
    y, _, stats, cnt = ops.conv_fwd(nhwc(x).cuda(), ops.pack_conv(w.cuda()), C, 3, 1, bias=bias.cuda(), want_stats=True)
    mean, rstd, _, _ = ops.bn_finalize(stats, cnt, torch.ones(C).cuda(), torch.zeros(C).cuda())
    ref = x.double() + 100.0
    v = (x + 100.0).double().var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(v + 1e-5)) < 1e-4


def test_conv_shuffle_and_clamp_stores(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 12, 12, generator=g)
    w = torch.randn(64, 16, 3, 3, generator=g) / 12.0
    ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), None, 1, 1), 2)
    y, _, _, _ = ops.conv_fwd(nhwc(x).cuda(), ops.pack_conv(w.cuda()), 64, 3, 1, out_mode=ops.OUT_SHUFFLE)
    assert tuple(y.shape) == (2, 24, 24, 16) and rel_err(nchw(y.cpu()), ref) < TOL
    # inverse store: unshuffle(conv) - feed the shuffled tensor through an identity 3x3
    w2 = torch.randn(3, 16, 9, 9, generator=g) / 36.0
    b2 = torch.randn(3, generator=g) * 0.5 + 0.5
    pre = F.conv2d(ref, w2.double(), b2.double(), 1, 4)
    yc, ypre, _, _ = ops.conv_fwd(y, ops.pack_conv(w2.cuda()), 3, 9, 1, bias=b2.cuda(), out_mode=ops.OUT_NCHW_CLAMP,
                                  want_pre=True)
    assert rel_err(ypre.cpu(), pre) < TOL
    assert torch.equal(yc.cpu(), ypre.cpu().clamp(0, 1))
    wi = torch.zeros(16, 16, 3, 3)
    for c in range(16):
        wi[c, c, 1, 1] = 1.0
    yu, _, _, _ = ops.conv_fwd(y, ops.pack_conv(wi.cuda()), 16, 3, 1, out_mode=ops.OUT_UNSHUFFLE)
    assert rel_err(nchw(yu.cpu()), F.pixel_unshuffle(ref, 2)) < TOL


@pytest.mark.parametrize("case", [(16, 24, 24), (4, 48, 48), (5, 42, 48)])
def test_conv_shuffle_store_nsplit_kernel(ops, case, monkeypatch):
    """The up-sampling blocks' convs (model.py:157-161: 64 -> 256, PixelShuffle(2), PReLU applied by the consumer) at sizes the N-split
    kernel takes (csrc/conv_nsplit.hip, OUT_SHUFFLE store; bias, PReLU on the input as the second block sees it): against torch in
    fp64, and against the general kernel's store of the same layer."""
    from srganst import _abi
    B, H, W = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(256, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(256, generator=g) * 0.1
    slope = torch.tensor([0.25])
    assert _abi.lib().sst_conv_ns_supported(B, H, W, 64, 256, 3, 1, ops.OUT_SHUFFLE)
    ref = F.pixel_shuffle(F.conv2d(F.prelu(x.double(), slope.double()), w.double(), b.double(), 1, 1), 2)
    wp = ops.pack_conv(w.cuda())
    kw = dict(bias=b.cuda(), in_slope=slope.cuda(), in_act=ops.ACT_SLOPE, out_mode=ops.OUT_SHUFFLE)
    y, _, _, _ = ops.conv_fwd(nhwc(x).cuda(), wp, 256, 3, 1, **kw)
    assert tuple(y.shape) == (B, 2 * H, 2 * W, 64) and rel_err(nchw(y.cpu()), ref) < TOL
    monkeypatch.setattr(ops, "CONV_NS", False)
    y0, _, _, _ = ops.conv_fwd(nhwc(x).cuda(), wp, 256, 3, 1, **kw)
    assert rel_err(y, y0) < 1e-5


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64, 3), (1, 13, 9, 8, 8, 3), (1, 16, 16, 64, 256, 3), (1, 16, 16, 256, 64, 3),
                                  (1, 20, 12, 64, 3, 9)])
def test_conv_dgrad_via_mode1_pack(ops, case):
    B, H, W, Cin, Cout, k = case
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    dy = torch.randn(B, Cout, H, W, generator=g)
    F.conv2d(x, w.double(), None, 1, k // 2).backward(dy.double())
    wd = ops.pack_conv(w.cuda(), 1)
    dx, _, _, _ = ops.conv_fwd(nhwc(dy).cuda(), wd, Cin, k, 1)
    assert rel_err(nchw(dx.cpu()), x.grad) < TOL


def test_conv_wgrad_deferred_slab_reduces(ops):
    """conv_wgrad(defer_reduce=jobs) + ONE wgrad_reduce_flush for several layers of different shapes (sst_wgrad_reduce_multi) against the
    per-layer reduce: bit-identical dW, with and without accumulation; shapes that write dW directly leave no job behind."""
    g = torch.Generator().manual_seed(15)
    cases = [(8, 48, 48, 64, 128, 3, 1), (8, 48, 48, 128, 128, 3, 2), (4, 24, 24, 64, 64, 3, 1), (2, 13, 9, 8, 8, 3, 1),
             (2, 24, 24, 3, 64, 9, 1), (8, 12, 12, 512, 512, 3, 2), (4, 96, 96, 3, 64, 3, 1)]
    jobs, pairs = [], []
    for (B, H, W, Cin, Cout, k, s) in cases:
        x = torch.randn(B, H, W, Cin, generator=g).cuda()
        ho, wo = ops.conv_out_hw(H, W, k, s)
        dy = torch.randn(B, ho, wo, Cout, generator=g).cuda()
        base = torch.randn(Cout, Cin, k, k, generator=g).cuda()
        for acc in (False, True):
            ref = base.clone()
            ops.conv_wgrad(x, dy, ref, k, s, accumulate=acc)
            out = base.clone()
            ops.conv_wgrad(x, dy, out, k, s, accumulate=acc, defer_reduce=jobs)
            pairs.append((ref, out, (B, H, W, Cin, Cout, k, s, acc)))
    assert 6 <= len(jobs) <= 24                       # some shapes write dW directly
    ops.wgrad_reduce_flush(jobs)
    assert not jobs
    torch.cuda.synchronize()
    for ref, out, c in pairs:
        assert torch.equal(ref, out), c


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64, 3, 1), (1, 13, 9, 8, 8, 3, 1), (2, 16, 16, 64, 256, 3, 1),
                                  (2, 24, 24, 64, 128, 3, 2), (2, 24, 24, 3, 64, 9, 1), (1, 32, 24, 64, 3, 9, 1),
                                  (1, 11, 7, 3, 64, 3, 1)])
def test_conv_wgrad(ops, case):
    B, H, W, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64, requires_grad=True)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xin = F.leaky_relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.2)
    y = F.conv2d(xin, w, None, s, k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dw = torch.full((Cout, Cin, k, k), 7.0).cuda()
    ops.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, k, s, in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2,
                   in_act=ops.ACT_SLOPE)
    assert rel_err(dw.cpu(), w.grad) < TOL
    ops.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, k, s, in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2,
                   in_act=ops.ACT_SLOPE, accumulate=True)
    assert rel_err(dw.cpu(), 2 * w.grad) < TOL


@pytest.mark.parametrize("C", [8, 64, 256])
def test_bn_prelu_backward_chain(ops, C):
    """conv-out y -> BN(train) -> PReLU -> (upstream g): dy, dgamma, dbeta, dslope vs autograd."""
    B, H, W = 2, 12, 12
    g = torch.Generator().manual_seed(6 + C)
    y = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    slope = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
    up = torch.randn(B, C, H, W, generator=g)
    up2 = torch.randn(B, C, H, W, generator=g)
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    F.prelu(z, slope).backward((up + up2).double())
    yf = y.detach().float()
    mean = yf.double().mean(dim=(0, 2, 3))
    var = yf.double().var(dim=(0, 2, 3), unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    scale = (gamma.detach() * rstd).float().cuda()
    shift = (beta.detach() - mean * gamma.detach() * rstd).float().cuda()
    yd, gd, g2d, sl = nhwc(yf).cuda(), nhwc(up).cuda(), nhwc(up2).cuda(), slope.detach().float().cuda()
    part = ops.bwd_reduce(gd, yd, g2=g2d, scale=scale, shift=shift, slope=sl, act=1)
    dgamma, dbeta, dslope = torch.empty(C).cuda(), torch.empty(C).cuda(), torch.empty(1).cuda()
    cA, cB, cC = ops.bwd_finalize(part, B * H * W, mean.float().cuda(), rstd.float().cuda(), gamma.detach().float().cuda(),
                                  dgamma, dbeta, dslope)
    dy = ops.bwd_apply(gd, yd, g2=g2d, scale=scale, shift=shift, slope=sl, act=1, cA=cA, cB=cB, cC=cC)
    assert rel_err(nchw(dy.cpu()), y.grad) < 1e-4
    assert rel_err(dgamma.cpu(), gamma.grad) < 1e-4
    assert rel_err(dbeta.cpu(), beta.grad) < 1e-4
    assert rel_err(dslope.cpu(), slope.grad) < 1e-4


def test_bn_residual_and_add(ops):
    g = torch.Generator().manual_seed(9)
    y, res = torch.randn(2, 8, 8, 64, generator=g), torch.randn(2, 8, 8, 64, generator=g)
    sc, sh = torch.randn(64, generator=g), torch.randn(64, generator=g)
    sl = torch.tensor([0.3])
    out = ops.bn_residual(y.cuda(), sc.cuda(), sh.cuda(), res.cuda(), sl.cuda())
    ref = y * sc + sh + torch.where(res > 0, res, res * 0.3)
    assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    out = ops.bn_residual(y.cuda(), sc.cuda(), sh.cuda(), res.cuda())
    assert torch.allclose(out.cpu(), y * sc + sh + res, rtol=1e-6, atol=1e-6)
    assert torch.equal(ops.add(y.cuda(), res.cuda()).cpu(), y + res)


@pytest.mark.parametrize("case", [(2, 24, 24, 64), (1, 13, 20, 8), (1, 96, 96, 64), (1, 7, 100, 16)])
def test_wgrad_c3_both_kinds(ops, case):
    """9x9 weight gradients with the (kx, 3ch) -> N folding, against autograd (conv3: C->3 with PReLU'd input; conv1: 3->C)."""
    B, H, W, C = case
    g = torch.Generator().manual_seed(31)
    # conv3: y = conv9x9(prelu(u), w3)
    u = torch.randn(B, C, H, W, generator=g)
    w3 = torch.randn(3, C, 9, 9, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, 3, H, W, generator=g)
    F.conv2d(F.prelu(u.double(), torch.tensor([0.25], dtype=torch.float64)), w3, None, 1, 4).backward(dy.double())
    dw = torch.zeros(3, C, 9, 9).cuda()
    ops.wgrad_c3(nhwc(u).cuda(), nhwc(dy).cuda(), dw, 0, in_slope=torch.tensor([0.25]).cuda(), in_act=ops.ACT_SLOPE)
    assert rel_err(dw.cpu(), w3.grad) < TOL
    # conv1: y = conv9x9(x3, w1)
    x3 = torch.randn(B, 3, H, W, generator=g)
    w1 = torch.randn(C, 3, 9, 9, generator=g, dtype=torch.float64, requires_grad=True)
    dz = torch.randn(B, C, H, W, generator=g)
    F.conv2d(x3.double(), w1, None, 1, 4).backward(dz.double())
    dw1 = torch.zeros(C, 3, 9, 9).cuda()
    ops.wgrad_c3(nhwc(dz).cuda(), nhwc(x3).cuda(), dw1, 1)
    assert rel_err(dw1.cpu(), w1.grad) < TOL


@pytest.mark.parametrize("case", [(2, 24, 24, 64), (1, 13, 37, 8), (1, 96, 96, 64), (2, 9, 5, 16)])
def test_conv9_folded_forward_kernels(ops, case):
    """conv1-type (3->C), conv3 data-gradient (mode 1) and conv3 forward (C->3, NCHW + clamp) with the (kx,3ch) folding."""
    B, H, W, C = case
    g = torch.Generator().manual_seed(41)
    x3 = torch.randn(B, 3, H, W, generator=g)
    w1 = torch.randn(C, 3, 9, 9, generator=g) / 15.6
    b1 = torch.randn(C, generator=g)
    ref = F.conv2d(x3.double(), w1.double(), b1.double(), 1, 4)
    y = ops.conv9_c3_fwd(nhwc(x3).cuda(), w1.cuda(), 0, bias=b1.cuda())
    assert rel_err(nchw(y.cpu()), ref) < TOL
    # conv3 forward + its data-gradient
    u = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w3 = torch.randn(3, C, 9, 9, generator=g) / (C * 81) ** 0.5
    b3 = torch.rand(3, generator=g)
    pre = F.conv2d(F.prelu(u, torch.tensor([0.25], dtype=torch.float64)), w3.double(), b3.double(), 1, 4)
    sr, sr_pre = ops.conv9_to3_fwd(nhwc(u.detach().float()).cuda(), w3.cuda(), bias=b3.cuda(), in_slope=torch.tensor([0.25]).cuda(),
                                   in_act=ops.ACT_SLOPE, want_pre=True)
    assert rel_err(sr_pre.cpu(), pre.detach()) < TOL
    assert torch.equal(sr.cpu(), sr_pre.cpu().clamp(0, 1))
    p = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, 3, H, W, generator=g)
    F.conv2d(p, w3.double(), None, 1, 4).backward(dy.double())
    dp = ops.conv9_c3_fwd(nhwc(dy).cuda(), w3.cuda(), 1)
    assert rel_err(nchw(dp.cpu()), p.grad) < TOL


@pytest.mark.parametrize("case", [(2, 48, 48, 64, 256, 1), (1, 24, 24, 64, 128, 2), (2, 13, 21, 8, 96, 1), (1, 16, 16, 128, 64, 1),
                                  (1, 9, 9, 64, 64, 2), (1, 12, 12, 256, 512, 1)])
def test_conv_big_tile_kernel(ops, case, monkeypatch):
    """The 64x64-tile kernel (forced with SST_CONV_BIG=1) against fp64 conv2d: plain / prologue+stats+residual / shuffle /
    data-gradient with backward partials."""
    monkeypatch.setenv("SST_CONV_BIG", "1")
    monkeypatch.setenv("SST_CONV_PIPE", "0")        # keep the pipelined kernel (tests/test_conv_pipe_gpu.py) off these shapes
    from srganst import _abi
    n_before = _abi.lib().sst_debug_big_tile_launches()
    B, H, W, Cin, Cout, s = case
    g = torch.Generator().manual_seed(61)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xin = F.leaky_relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.2)
    ref = F.conv2d(xin, w.double(), bias.double(), s, 1)
    res = torch.randn(ref.shape, generator=g)
    wp = ops.pack_conv(w.cuda())
    y, _, stats, cnt = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, s, bias=bias.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                                    in_slope_const=0.2, in_act=ops.ACT_SLOPE, residual=nhwc(res).cuda(), want_stats=True)
    full = ref + res.double()
    assert rel_err(nchw(y.cpu()), full) < TOL
    mean, rstd, _, _ = ops.bn_finalize(stats, cnt, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda())
    assert rel_err(mean.cpu(), full.mean(dim=(0, 2, 3))) < 1e-4
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(full.var(dim=(0, 2, 3), unbiased=False) + 1e-5)) < 1e-4
    if s == 1 and Cout % 4 == 0:
        ysh = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, 1, out_mode=ops.OUT_SHUFFLE)[0]
        assert rel_err(nchw(ysh.cpu()), F.pixel_shuffle(F.conv2d(x.double(), w.double(), None, 1, 1), 2)) < TOL
        # data-gradient + backward partials against a saved tensor
        dy = torch.randn(B, Cout, H, W, generator=g)
        ysave = torch.randn(B, Cin, H, W, generator=g)
        gref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), 1, 1)
        gd, part = ops.conv_dgrad_bwdstats(nhwc(dy).cuda(), ops.pack_conv(w.cuda(), 1), Cin, 3, nhwc(ysave).cuda(),
                                           epi_scale=sc.cuda(), epi_shift=sh.cuda(), epi_slope_const=0.2, epi_act=1)
        assert rel_err(nchw(gd.cpu()), gref) < TOL
        z = ysave.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        gz = torch.where(z > 0, gref, gref * 0.2)
        tot = part.sum(dim=0).cpu()
        assert rel_err(tot[0], gz.sum(dim=(0, 2, 3))) < 1e-4
        assert rel_err(tot[1], (gz * ysave.double()).sum(dim=(0, 2, 3))) < 1e-4
        assert rel_err(tot[2], (gref * z.clamp(max=0)).sum(dim=(0, 2, 3))) < 1e-4
    else:
        xg = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        yy = F.conv2d(xg, w.double(), None, 2, 1)
        dy = torch.randn(yy.shape, generator=g)
        yy.backward(dy.double())
        dx = ops.conv_s2_dgrad(nhwc(dy).cuda(), ops.pack_conv_s2_dgrad(w.cuda()), H, W, Cin)
        assert rel_err(nchw(dx.cpu()), xg.grad) < TOL
    assert _abi.lib().sst_debug_big_tile_launches() >= n_before + 2      # the 64x64 kernel really ran


@pytest.mark.parametrize("case", [(2, 24, 24, 64, "9"), (2, 24, 24, 64, "3"), (1, 48, 48, 64, "9"), (1, 48, 48, 32, "3"),
                                  (3, 12, 12, 64, "9"), (1, 12, 12, 16, "3"), (2, 18, 8, 64, "9")])
def test_conv_band_kernel(ops, case, monkeypatch):
    """The band kernel (csrc/conv_band.hip; NB = 9 or 3 pixel blocks per band forced with SST_CONV_BAND) against fp64
    conv2d and against the general kernel: forward with prologue + bias + residual + BatchNorm statistics, the plain
    data-gradient with backward partials, and the fused BatchNorm-backward stage (apply on load, dy side output)."""
    B, H, W, Cout, nb = case
    Cin = 64
    from srganst import _abi
    monkeypatch.setenv("SST_CONV_BAND", nb)
    assert _abi.lib().sst_conv_stat_tiles(B, H, W, Cin, Cout, 3, 1) == B * H * W // (16 * int(nb))
    n_before = _abi.lib().sst_debug_band_launches()
    g = torch.Generator().manual_seed(71)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    slope = torch.tensor([0.25])
    xin = F.prelu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), slope.double())
    res = torch.randn(B, Cout, H, W, generator=g)
    full = F.conv2d(xin, w.double(), bias.double(), 1, 1) + res.double()
    wp = ops.pack_conv(w.cuda())
    y, _, stats, cnt = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, 1, bias=bias.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                                    in_slope=slope.cuda(), in_act=ops.ACT_SLOPE, residual=nhwc(res).cuda(), want_stats=True)
    assert rel_err(nchw(y.cpu()), full) < TOL
    assert stats.shape[0] == B * H * W // (16 * int(nb)) and float(cnt.sum()) == B * H * W
    mean, rstd, _, _ = ops.bn_finalize(stats, cnt, torch.ones(Cout).cuda(), torch.zeros(Cout).cuda())
    assert rel_err(mean.cpu(), full.mean(dim=(0, 2, 3))) < 1e-4
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(full.var(dim=(0, 2, 3), unbiased=False) + 1e-5)) < 1e-4

    # data-gradient of a (Cout -> 64)... roles: the conv below maps 64 "dy" channels to Cout "dx" channels
    wt = torch.randn(Cin, Cout, 3, 3, generator=g) / 24.0            # a conv Cout -> 64 whose dgrad has 64 inputs
    dy = torch.randn(B, Cin, H, W, generator=g)
    gref = torch.nn.grad.conv2d_input((B, Cout, H, W), wt.double(), dy.double(), 1, 1)
    ysave = torch.randn(B, Cout, H, W, generator=g)
    esc, esh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    wd = ops.pack_conv(wt.cuda(), 1)
    gd, part = ops.conv_dgrad_bwdstats(nhwc(dy).cuda(), wd, Cout, 3, nhwc(ysave).cuda(), residual=nhwc(res).cuda(),
                                       epi_scale=esc.cuda(), epi_shift=esh.cuda(), epi_slope=slope.cuda(), epi_act=1)
    gfull = gref + res.double()
    assert rel_err(nchw(gd.cpu()), gfull) < TOL
    z = ysave.double() * esc.double().view(1, -1, 1, 1) + esh.double().view(1, -1, 1, 1)
    gz = torch.where(z > 0, gfull, gfull * 0.25)
    tot = part.sum(dim=0).cpu()
    assert part.shape[0] == stats.shape[0]
    assert rel_err(tot[0], gz.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[1], (gz * ysave.double()).sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[2], (gfull * z.clamp(max=0)).sum(dim=(0, 2, 3))) < 1e-4

    # fused BatchNorm-backward stage: must equal the general kernel bit for bit in dy, and to rounding in the conv
    y2 = torch.randn(B, Cin, H, W, generator=g)
    cA, cB, cC = (torch.randn(Cin, generator=g) for _ in range(3))
    args = dict(cA=cA.cuda(), cB=cB.cuda(), cC=cC.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope=slope.cuda(), in_act=1,
                residual=nhwc(res).cuda(), epi_y=nhwc(ysave).cuda(), epi_scale=esc.cuda(), epi_shift=esh.cuda(),
                epi_slope=slope.cuda(), epi_act=1)
    out_b, dy_b, part_b = ops.conv_dgrad_fused(nhwc(dy).cuda(), nhwc(y2).cuda(), wd, Cout, 3, **args)
    assert _abi.lib().sst_debug_band_launches() == n_before + 3           # the band kernel really ran, every time
    monkeypatch.setenv("SST_CONV_BAND", "0")
    out_g, dy_g, part_g = ops.conv_dgrad_fused(nhwc(dy).cuda(), nhwc(y2).cuda(), wd, Cout, 3, **args)
    assert _abi.lib().sst_debug_band_launches() == n_before + 3
    assert torch.equal(dy_b, dy_g)
    assert rel_err(out_b.cpu(), out_g.cpu()) < TOL
    assert rel_err(part_b.sum(dim=0).cpu(), part_g.sum(dim=0).cpu()) < 1e-4
    z2 = y2.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    gz2 = torch.where(z2 > 0, dy.double(), dy.double() * 0.25)
    dyr = cA.double().view(1, -1, 1, 1) * gz2 + cB.double().view(1, -1, 1, 1) * y2.double() + cC.double().view(1, -1, 1, 1)
    assert rel_err(nchw(dy_b.cpu()), dyr) < TOL
    oref = torch.nn.grad.conv2d_input((B, Cout, H, W), wt.double(), dyr, 1, 1) + res.double()
    assert rel_err(nchw(out_b.cpu()), oref) < TOL


@pytest.mark.parametrize("case", [(3, 32, 48, 64, 64, 2, "2, 8"), (2, 24, 24, 128, 96, 2, "2, 6"), (5, 12, 12, 96, 160, 2, "2, 6"),
                                  (16, 12, 12, 512, 512, 2, "2, 6"), (1, 4, 16, 32, 32, 2, "2, 8"), (2, 96, 96, 64, 64, 2, "2, 8"),
                                  (2, 8, 16, 32, 32, 1, "4, 8"), (3, 16, 24, 64, 96, 1, "4, 8"), (2, 12, 12, 64, 32, 1, "4, 6"),
                                  (16, 12, 12, 256, 512, 1, "4, 6"), (1, 4, 8, 32, 64, 1, "4, 8"), (2, 48, 48, 64, 128, 1, "4, 8")])
def test_conv_wgrad_all_taps_tile_kernel(ops, case, monkeypatch):
    """The all-taps weight-gradient kernel for single layers (conv_wgrad_tile_kernel: 32 x 32 block x 9 taps per workgroup, its 8
    waves split the pixel tiles; stride 2 with 2x8 / 2x6 tiles, stride 1 with 4x8 / 4x6 tiles) against fp64 autograd and against
    the kernels it replaces (SST_WGRAD_S2=0 / SST_WGRAD_S1T=0): with the producer's BatchNorm affine + LeakyReLU applied to the input
    and without, writing and accumulating, interior and edge tiles."""
    from srganst import _abi
    B, H, W, Cin, Cout, stride, tile = case
    env = "SST_WGRAD_S2" if stride == 2 else "SST_WGRAD_S1T"
    monkeypatch.setenv(env, "1")                          # also below the work threshold of the automatic choice
    g = torch.Generator().manual_seed(83)
    name = _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, Cin, Cout, 3, stride, 1).decode()
    assert name == f"conv_wgrad_tile_kernel<{stride}, {tile}, 8>", name
    x = torch.randn(B, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    dy = torch.randn(B, Cout, H // stride, W // stride, generator=g)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    for affine in (True, False):
        w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        xin = F.leaky_relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.2) if affine else x.double()
        F.conv2d(xin, w, None, stride, 1).backward(dy.double())
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2, in_act=ops.ACT_SLOPE) if affine else {}
        dw = torch.full((Cout, Cin, 3, 3), 7.0).cuda()
        ops.conv_wgrad(xd, dyd, dw, 3, stride, **kw)
        assert rel_err(dw.cpu(), w.grad) < TOL
        ops.conv_wgrad(xd, dyd, dw, 3, stride, accumulate=True, **kw)
        assert rel_err(dw.cpu(), 2 * w.grad) < TOL
        monkeypatch.setenv(env, "0")
        assert "tile" not in _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, Cin, Cout, 3, stride, 1).decode()
        dw_old = torch.empty_like(dw)
        ops.conv_wgrad(xd, dyd, dw_old, 3, stride, **kw)
        monkeypatch.setenv(env, "1")
        assert rel_err(dw_old.cpu(), w.grad) < TOL
        dw2 = torch.empty_like(dw)
        ops.conv_wgrad(xd, dyd, dw2, 3, stride, **kw)
        dw3 = torch.empty_like(dw)
        ops.conv_wgrad(xd, dyd, dw3, 3, stride, **kw)
        assert torch.equal(dw2, dw3)                      # fixed-order sums: reproducible
    # a PReLU-style slope outside [0, 1] takes the select form of the activation
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.where(x.double() > 0, x.double(), -1.5 * x.double()), w, None, stride, 1).backward(dy.double())
    dw = torch.empty(Cout, Cin, 3, 3).cuda()
    ops.conv_wgrad(xd, dyd, dw, 3, stride, in_slope_const=-1.5, in_act=ops.ACT_SLOPE)
    assert rel_err(dw.cpu(), w.grad) < TOL


@pytest.mark.parametrize("case", [(2, 8, 32, 64), (3, 5, 64, 64), (1, 1, 32, 64), (16, 96, 96, 64), (2, 12, 96, 64)])
def test_conv_wgrad_3_channel_input_mfma_kernel(ops, case, monkeypatch):
    """Weight gradient of the discriminator's first layer (3 input channels, reference model.py:32) on the matrix cores
    (wgrad_k3c3_mfma_kernel: N = (ky, kx, ci) = 27 of 32 columns read straight from the raw patch) against fp64 autograd and against
    the VALU kernel it replaces, writing and accumulating; image borders on all four sides, partial last workgroup."""
    from srganst import _abi
    B, H, W, Cout = case
    assert _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, 3, Cout, 3, 1, 1).decode() == "wgrad_k3c3_mfma_kernel"
    g = torch.Generator().manual_seed(85)
    x = torch.randn(B, 3, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(dy.double())
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    dw = torch.full((Cout, 3, 3, 3), 7.0).cuda()
    ops.conv_wgrad(xd, dyd, dw, 3, 1)
    assert rel_err(dw.cpu(), w.grad) < TOL
    ops.conv_wgrad(xd, dyd, dw, 3, 1, accumulate=True)
    assert rel_err(dw.cpu(), 2 * w.grad) < TOL
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw2, 3, 1)
    dw3 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw3, 3, 1)
    assert torch.equal(dw2, dw3)
    monkeypatch.setenv("SST_WGRAD_NO_K3C3_MFMA", "1")
    assert _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, 3, Cout, 3, 1, 1).decode() == "wgrad_k3c3_kernel"
    dw_old = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw_old, 3, 1)
    assert rel_err(dw_old.cpu(), w.grad) < TOL


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64), (1, 48, 48, 64, 128), (3, 12, 12, 128, 64), (1, 9, 16, 64, 64), (2, 6, 8, 64, 64),
                                  (16, 24, 24, 64, 64)])
def test_conv_wgrad_band_kernel(ops, case, monkeypatch):
    """The all-taps weight-gradient kernel (3x3, stride 1, channels multiples of 64) against fp64 autograd and against the
    per-tap kernel (SST_WGRAD_BAND=0), single launch and grouped launch."""
    from srganst import _abi
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(81)
    monkeypatch.setenv("SST_WGRAD_BAND", "1")          # force it also below the work threshold of the automatic choice
    n0 = _abi.lib().sst_debug_wgrad_band_launches()
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xin = F.leaky_relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.2)
    y = F.conv2d(xin, w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2, in_act=ops.ACT_SLOPE)
    dw = torch.full((Cout, Cin, 3, 3), 7.0).cuda()
    ops.conv_wgrad(xd, dyd, dw, 3, 1, **kw)
    assert _abi.lib().sst_debug_wgrad_band_launches() == n0 + 1
    assert rel_err(dw.cpu(), w.grad) < TOL
    ops.conv_wgrad(xd, dyd, dw, 3, 1, accumulate=True, **kw)
    assert rel_err(dw.cpu(), 2 * w.grad) < TOL
    monkeypatch.setenv("SST_WGRAD_BAND", "0")
    dw_old = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw_old, 3, 1, **kw)
    assert _abi.lib().sst_debug_wgrad_band_launches() == n0 + 2
    monkeypatch.setenv("SST_WGRAD_BAND", "1")
    assert rel_err(dw_old.cpu(), w.grad) < TOL
    # grouped: three layers of this shape (different data) in one launch
    grp = ops.WgradGroup()
    outs, refs = [], []
    for i in range(3):
        xi = torch.randn(B, Cin, H, W, generator=g)
        wi = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        yi = F.conv2d(xi.double(), wi, None, 1, 1)
        dyi = torch.randn(yi.shape, generator=g)
        yi.backward(dyi.double())
        o = torch.empty(Cout, Cin, 3, 3).cuda()
        grp.add(nhwc(xi).cuda(), nhwc(dyi).cuda(), o, 3, 1)
        outs.append(o)
        refs.append(wi.grad)
    grp.run()
    assert _abi.lib().sst_debug_wgrad_band_launches() == n0 + 3
    for o, r in zip(outs, refs):
        assert rel_err(o.cpu(), r) < TOL


@pytest.mark.parametrize("case", [(2, 16, 24, 64, True), (1, 6, 10, 16, True), (2, 13, 9, 64, False), (1, 48, 48, 64, True), (3, 5, 7, 8, False),
                                  (1, 8, 8, 4, True)])
def test_act_bwd_with_partials(ops, case):
    """PReLU backward (+ inverse PixelShuffle) with bias / slope gradients from the same pass, against autograd in fp64."""
    B, H, W, C, unsh = case
    g = torch.Generator().manual_seed(91)
    y = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)     # activation input (shuffled layout)
    slope = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
    up = torch.randn(B, C, H, W, generator=g)
    up2 = torch.randn(B, C, H, W, generator=g)
    F.prelu(y, slope).backward((up + up2).double())
    dslope = torch.full((1,), 3.0).cuda()
    Cs = 4 * C if unsh else C
    dbias = torch.full((Cs,), 5.0).cuda()
    dy = ops.act_bwd(nhwc(up).cuda(), nhwc(y.detach().float()).cuda(), g2=nhwc(up2).cuda(), slope=slope.detach().float().cuda(),
                     dbias=dbias, dslope=dslope, unshuffle=unsh)
    ref = F.pixel_unshuffle(y.grad, 2) if unsh else y.grad
    assert rel_err(nchw(dy.cpu()), ref) < TOL
    assert rel_err(dbias.cpu(), ref.sum(dim=(0, 2, 3))) < 1e-5
    assert abs(float(dslope) - float(slope.grad)) < 1e-4 * max(1.0, abs(float(slope.grad)))


@pytest.mark.parametrize("case", [(2, 24, 20, 64), (1, 96, 96, 64), (3, 13, 9, 128), (2, 8, 8, 16)])
def test_conv_wgrad_three_channel_input(ops, case, monkeypatch):
    """Weight gradient of a 3x3 stride-1 conv with a 3-channel input (Discriminator.features[0]): dedicated VALU kernel vs
    fp64 autograd and vs the general kernel."""
    B, H, W, Cout = case
    g = torch.Generator().manual_seed(95)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(Cout, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    from srganst import _abi
    monkeypatch.setenv("SST_WGRAD_NO_K3C3_MFMA", "1")     # the MFMA form of this layer has its own test
    assert _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, 3, Cout, 3, 1, 1) == b"wgrad_k3c3_kernel"
    dw = torch.full((Cout, 3, 3, 3), 7.0).cuda()
    ops.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, 3, 1)
    assert rel_err(dw.cpu(), w.grad) < TOL
    ops.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, 3, 1, accumulate=True)
    assert rel_err(dw.cpu(), 2 * w.grad) < TOL
    monkeypatch.setenv("SST_WGRAD_NO_K3C3", "1")
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw2, 3, 1)
    assert rel_err(dw2.cpu(), w.grad) < TOL


@pytest.mark.parametrize("case", [(2, 24, 20, 64), (1, 96, 96, 64), (3, 13, 9, 128), (2, 8, 8, 16)])
def test_conv_fwd_three_channel_input(ops, case, monkeypatch):
    """3x3 stride-1 conv from a 3-channel image with bias (Discriminator.features[0]): dedicated VALU kernel vs fp64 conv2d and
    vs the general kernel."""
    B, H, W, Cout = case
    from srganst import _abi
    g = torch.Generator().manual_seed(97)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(Cout, 3, 3, 3, generator=g) / 5.0
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    monkeypatch.setenv("SST_NO_C3IN_MFMA", "1")           # the MFMA form of this layer has its own test
    assert _abi.lib().sst_conv_kernel_name(B, H, W, 3, Cout, 3, 1, 0, 0) == b"conv3_c3in_kernel"
    wp = ops.pack_conv(w.cuda())
    y = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, 1, bias=b.cuda())[0]
    assert rel_err(nchw(y.cpu()), ref) < TOL
    monkeypatch.setenv("SST_NO_C3IN", "1")
    y2 = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, 1, bias=b.cuda())[0]
    assert rel_err(y2.cpu(), y.cpu()) < TOL


@pytest.mark.parametrize("case", [(2, 8, 32), (3, 5, 64), (1, 1, 32), (16, 96, 96), (2, 12, 96), (1, 7, 192)])
def test_conv_fwd_three_channel_input_mfma_kernel(ops, case, monkeypatch):
    """The discriminator's first layer (3 -> 64, bias, reference model.py:32) on the matrix cores (conv3_c3in_mfma_kernel: K = 3 rows
    x 10 columns read straight from the raw patch, output written as whole 256-B pixel rows) vs fp64 conv2d and vs the VALU kernel,
    with and without bias; image borders on all four sides, partial last workgroup."""
    from srganst import _abi
    B, H, W = case
    g = torch.Generator().manual_seed(98)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5.0
    b = torch.randn(64, generator=g)
    assert _abi.lib().sst_conv_kernel_name(B, H, W, 3, 64, 3, 1, 0, 0) == b"conv3_c3in_mfma_kernel"
    wp = ops.pack_conv(w.cuda())
    y = ops.conv_fwd(nhwc(x).cuda(), wp, 64, 3, 1, bias=b.cuda())[0]
    assert rel_err(nchw(y.cpu()), F.conv2d(x.double(), w.double(), b.double(), 1, 1)) < TOL
    y0 = ops.conv_fwd(nhwc(x).cuda(), wp, 64, 3, 1)[0]
    assert rel_err(nchw(y0.cpu()), F.conv2d(x.double(), w.double(), None, 1, 1)) < TOL
    monkeypatch.setenv("SST_NO_C3IN_MFMA", "1")
    y2 = ops.conv_fwd(nhwc(x).cuda(), wp, 64, 3, 1, bias=b.cuda())[0]
    assert rel_err(y2.cpu(), y.cpu()) < TOL


@pytest.mark.parametrize("case", [(2, 8, 32), (16, 96, 96), (3, 5, 20), (1, 7, 192), (2, 3, 14), (5, 1, 16)])
def test_conv_64_to_3_channels_kernel(ops, case, monkeypatch):
    """3x3 conv from 64 channels to 3 with kx folded into the MFMA columns (conv3_to3_kernel): as the data-gradient of the
    discriminator's first layer (reference model.py:32, mode-1 packed weights) against autograd in fp64, as a plain forward conv
    (mode-0 pack) against conv2d, and against the general kernel (SST_NO_TO3); borders, widths that are not multiples of 16."""
    from srganst import _abi
    B, H, W = case
    g = torch.Generator().manual_seed(101)
    assert _abi.lib().sst_conv_kernel_name(B, H, W, 64, 3, 3, 1, 0, 0) == b"conv3_to3_kernel"
    x = torch.randn(B, 3, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5.0
    dy = torch.randn(B, 64, H, W, generator=g)
    F.conv2d(x, w.double(), None, 1, 1).backward(dy.double())
    dyd = nhwc(dy).cuda()
    wd = ops.pack_conv(w.cuda(), 1)
    dx = ops.conv_fwd(dyd, wd, 3, 3, 1)[0]
    assert tuple(dx.shape) == (B, H, W, 3) and rel_err(nchw(dx.cpu()), x.grad) < TOL
    w2 = torch.randn(3, 64, 3, 3, generator=g) / 24.0                        # the same kernel as a forward conv
    y = ops.conv_fwd(dyd, ops.pack_conv(w2.cuda()), 3, 3, 1)[0]
    assert rel_err(nchw(y.cpu()), F.conv2d(dy.double(), w2.double(), None, 1, 1)) < TOL
    monkeypatch.setenv("SST_NO_TO3", "1")
    assert _abi.lib().sst_conv_kernel_name(B, H, W, 64, 3, 3, 1, 0, 0) != b"conv3_to3_kernel"
    dx_old = ops.conv_fwd(dyd, wd, 3, 3, 1)[0]
    assert rel_err(dx_old.cpu(), dx.cpu()) < TOL


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64), (1, 12, 12, 128, 64), (2, 9, 13, 64, 128), (1, 6, 6, 256, 256)])
def test_conv_s2_dgrad_fused_stage(ops, case):
    """BatchNorm-backward stage around the stride-2 data-gradient (apply on load, dy side output, partial sums of the result
    against the previous layer's output) vs explicit composition in fp64."""
    B, H, W, Cin, Cout = case
    g_ = torch.Generator().manual_seed(99)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / (Cin * 9) ** 0.5
    gup = torch.randn(B, Cout, ho, wo, generator=g_)              # gradient w.r.t. LeakyReLU(BN(y2))
    y2 = torch.randn(B, Cout, ho, wo, generator=g_)
    sc, sh = torch.rand(Cout, generator=g_) + 0.5, torch.randn(Cout, generator=g_) * 0.3
    cA, cB, cC = (torch.randn(Cout, generator=g_) for _ in range(3))
    v = lambda t: t.double().view(1, -1, 1, 1)
    z = y2.double() * v(sc) + v(sh)
    gz = torch.where(z > 0, gup.double(), gup.double() * 0.2)
    dy_ref = v(cA) * gz + v(cB) * y2.double() + v(cC)
    dx_ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy_ref, 2, 1)
    yprev = torch.randn(B, Cin, H, W, generator=g_)
    esc, esh = torch.rand(Cin, generator=g_) + 0.5, torch.randn(Cin, generator=g_) * 0.3
    dx, dy, part = ops.conv_s2_dgrad_fused(nhwc(gup).cuda(), nhwc(y2).cuda(), ops.pack_conv_s2_dgrad(w.cuda()), H, W, Cin,
                                           cA=cA.cuda(), cB=cB.cuda(), cC=cC.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                                           in_slope_const=0.2, in_act=1, epi_y=nhwc(yprev).cuda(), epi_scale=esc.cuda(),
                                           epi_shift=esh.cuda(), epi_slope_const=0.2, epi_act=1)
    assert rel_err(nchw(dy.cpu()), dy_ref) < TOL
    assert rel_err(nchw(dx.cpu()), dx_ref) < TOL
    zp = yprev.double() * esc.double().view(1, -1, 1, 1) + esh.double().view(1, -1, 1, 1)
    gzp = torch.where(zp > 0, dx_ref, dx_ref * 0.2)
    tot = part.sum(dim=0).cpu()
    assert rel_err(tot[0], gzp.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[1], (gzp * yprev.double()).sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[2], (dx_ref * zp.clamp(max=0)).sum(dim=(0, 2, 3))) < 1e-4


def test_multi_pack_9x9_modes_equal_standalone_packers():
    """PackPlan modes 2..4 (the (kx, 3ch)-folded layouts of the 9x9 convs inside the multi-tensor pack launch) against the
    stand-alone pack kernels: bit-identical buffers, next to ordinary forward / data-gradient jobs in the same launch."""
    from srganst import _abi, ops
    g = torch.Generator().manual_seed(77)
    w1 = torch.randn(64, 3, 9, 9, generator=g).cuda()           # conv1
    w3 = torch.randn(3, 64, 9, 9, generator=g).cuda()           # conv3
    wt = torch.randn(64, 64, 3, 3, generator=g).cuda()          # a trunk conv riding in the same launch
    out = ops.PackPlan([wt, w1, w3, w3, wt], [ops.PACK_FWD, ops.PACK_C3_FWD, ops.PACK_TO3, ops.PACK_C3_DGRAD, ops.PACK_DGRAD]).run()
    lib = _abi.lib()

    def c3(w, mode):
        o = w.shape[1] if mode else w.shape[0]
        wp = torch.empty(lib.sst_conv9_c3_packed_floats(o), device="cuda")
        _abi.check(lib.sst_conv9_c3_pack(_abi.ptr(w), _abi.ptr(wp), w.shape[0], w.shape[1], mode, _abi.stream_ptr()), "pack")
        return wp

    t3 = torch.empty(lib.sst_conv9_to3_packed_floats(64), device="cuda")
    _abi.check(lib.sst_conv9_to3_pack(_abi.ptr(w3), _abi.ptr(t3), 64, _abi.stream_ptr()), "pack")
    assert torch.equal(out[1], c3(w1, 0))
    assert torch.equal(out[2], t3)
    assert torch.equal(out[3], c3(w3, 1))
    assert torch.equal(out[0], ops.pack_conv(wt, 0)) and torch.equal(out[4], ops.pack_conv(wt, 1))

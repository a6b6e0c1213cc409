"""GPU parity of the persistent, software-pipelined 3x3 conv kernel (csrc/conv_pipe.hip, sst_conv_pipe_fwd) against torch fp64:
every tile width (8 / 4 / 2 pixels), stride 1 and 2, tiles that straddle images of the tall batch image, a partial last tile,
K split over workgroups (workspace slabs + reduce kernel), fused input affine + LeakyReLU, bias, BatchNorm forward statistics
and the BatchNorm/activation backward partials of the data-gradient form.  Replaces cuDNN/oneDNN behind the discriminator's
convs (reference model.py:30-59).  Tolerance: 2e-5 norm-wise against fp64 (fp32 MFMA = exact fp32 FMA chain)."""
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


# (B, H, W, Cin, Cout, stride): expected (tile width, K-split)
CASES = {
    (16, 12, 12, 256, 512, 1): (4, 1),     # discriminator layer 7: 8-row tiles straddle the 12-row images
    (16, 12, 12, 512, 512, 2): (2, 4),     # layer 8: 6x6 outputs, 16-row tiles cross up to 3 image boundaries, K split 4 ways
    (4, 24, 24, 128, 256, 1): (8, 2),      # K split 2 ways
    (5, 22, 24, 64, 128, 1): (8, 1),       # Ho % 4 != 0: straddling 8x4 tiles, partial last tile (110 rows)
    (6, 14, 12, 64, 256, 1): (4, 1),       # straddle + partial last tile, 4-wide tiles
    (8, 48, 48, 128, 128, 2): (8, 2),      # layer 4
    (16, 24, 24, 256, 256, 2): (4, 2),     # layer 6
    (9, 20, 12, 64, 512, 2): (2, 1),       # 2-wide tiles, partial last tile (90 rows), stride 2
    (16, 6, 6, 128, 512, 1): (2, 2),       # 2-wide tiles, stride 1
    (4, 48, 48, 64, 128, 1): (8, 1),       # layer 3 shape
}


def test_plan_and_dispatch(ops):
    from srganst import _abi
    L = _abi.lib()
    for (B, H, W, Cin, Cout, s), (tw, ks) in CASES.items():
        assert L.sst_conv_pipe_supported(B, H, W, Cin, Cout, 3, s) == tw, (B, H, W, Cin, Cout, s)
        ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
        th = 32 // tw
        n_mt = (wo // tw) * ((B * ho + th - 1) // th)
        assert L.sst_conv_pipe_stat_tiles(B, H, W, Cin, Cout, 3, s) == n_mt
        assert L.sst_conv_pipe_ws_floats(B, H, W, Cin, Cout, 3, s) == (ks * n_mt * (Cout // 32) * 1024 if ks > 1 else 0)
    # not taken: the trunk shape (band kernel), channel counts off the 64 / 32 grid, odd sizes at stride 2, too few tiles
    for shp in [(16, 24, 24, 64, 64, 3, 1), (2, 24, 24, 48, 64, 3, 1), (2, 24, 24, 64, 48, 3, 1), (16, 13, 13, 64, 64, 3, 2),
                (1, 8, 8, 64, 64, 3, 1), (2, 24, 24, 64, 64, 9, 1)]:
        assert L.sst_conv_pipe_supported(*shp) == 0, shp


@pytest.mark.parametrize("case", list(CASES))
def test_conv_pipe_forward_affine_stats(ops, case):
    B, H, W, Cin, Cout, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    wp = ops.pack_conv(w.cuda())
    # plain + bias
    ref = F.conv2d(x.double(), w.double(), bias.double(), s, 1)
    y = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, s, bias=bias.cuda())[0]
    assert rel_err(nchw(y.cpu()), ref) < TOL
    # producer's BatchNorm affine + LeakyReLU applied while staging (zero padding stays zero), BatchNorm statistics of the output
    xin = F.leaky_relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), 0.2)
    ref = F.conv2d(xin, w.double(), None, s, 1)
    y, _, stats, cnt = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, s, in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2,
                                    in_act=ops.ACT_SLOPE, want_stats=True)
    assert rel_err(nchw(y.cpu()), ref) < TOL
    assert stats.shape[0] == cnt.shape[0] and float(cnt.sum()) == B * ref.shape[2] * ref.shape[3]
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    mean, rstd, scale, shift = ops.bn_finalize(stats, cnt, gamma.cuda(), beta.cuda())
    m_ref, v_ref = ref.mean(dim=(0, 2, 3)), ref.var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(mean.cpu(), m_ref) < 1e-4
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(v_ref + 1e-5)) < TOL
    # same launch twice: bit-identical (fixed-order reductions, no atomics)
    y2, _, stats2, _ = ops.conv_fwd(nhwc(x).cuda(), wp, Cout, 3, s, in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope_const=0.2,
                                    in_act=ops.ACT_SLOPE, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(stats, stats2)


@pytest.mark.parametrize("case", [c for c in CASES if c[5] == 1])
def test_conv_pipe_dgrad_with_backward_partials(ops, case):
    """The stride-1 data-gradient form (weights packed with mode 1) + the BatchNorm/activation backward partial sums of its
    result against a saved tensor (ops.conv_dgrad_bwdstats), as disc_graph.backward uses it."""
    B, H, W, Cin, Cout, s = case
    g = torch.Generator().manual_seed(7 + sum(case))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    dy = torch.randn(B, Cout, H, W, generator=g)
    ysave = torch.randn(B, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    from srganst import _abi
    assert _abi.lib().sst_conv_pipe_supported(B, H, W, Cout, Cin, 3, 1), "the data-gradient shape must be taken as well"
    gref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), 1, 1)
    gd, part = ops.conv_dgrad_bwdstats(nhwc(dy).cuda(), ops.pack_conv(w.cuda(), 1), Cin, 3, nhwc(ysave).cuda(),
                                       epi_scale=sc.cuda(), epi_shift=sh.cuda(), epi_slope_const=0.2, epi_act=1)
    assert rel_err(nchw(gd.cpu()), gref) < TOL
    z = ysave.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    gz = torch.where(z > 0, gref, gref * 0.2)
    tot = part.double().sum(dim=0).cpu()
    assert rel_err(tot[0], gz.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[1], (gz * ysave.double()).sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(tot[2], (gref * z.clamp(max=0)).sum(dim=(0, 2, 3))) < 1e-4


def test_conv_pipe_equals_general_kernel(ops, monkeypatch):
    """Same conv through the general kernel (SST_CONV_PIPE=0): agreement to fp32 re-association, and the dev switch works."""
    B, H, W, Cin, Cout, s = 16, 12, 12, 256, 512, 1
    g = torch.Generator().manual_seed(5)
    x = nhwc(torch.randn(B, Cin, H, W, generator=g)).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 48.0).cuda()
    wp = ops.pack_conv(w)
    y1, _, st1, _ = ops.conv_fwd(x, wp, Cout, 3, s, want_stats=True)
    monkeypatch.setenv("SST_CONV_PIPE", "0")
    y0, _, st0, _ = ops.conv_fwd(x, wp, Cout, 3, s, want_stats=True)
    assert st0.shape[0] != st1.shape[0]                       # different statistics tilings: 96 (8x4, padded) vs 72 (tall image)
    assert rel_err(y1.cpu(), y0.cpu()) < 1e-5


# stride-2 data-gradient mode: (B, H, W, Cin, Cout) of the conv whose input gradient is computed -> (tile width, K-split)
S2D_CASES = {
    (16, 96, 96, 64, 64): (8, 1),          # discriminator layer 2
    (16, 48, 48, 128, 128): (8, 1),        # layer 4
    (16, 24, 24, 256, 256): (0, 0),        # layer 6 (12 x 12 class grid, would split K): measured slower, stays on conv_s2dgrad4_kernel
    (16, 24, 24, 128, 64): (4, 1),         # 12 x 12 class grid, 8-row tiles straddle images
    (16, 12, 12, 512, 512): (2, 4),        # layer 8: 6 x 6 class grid, 16-row tiles, K split 4 ways
    (5, 44, 48, 128, 64): (8, 1),          # class grid 22 x 24: straddling tiles, partial last tile
    (6, 28, 24, 256, 64): (4, 1),
    (9, 20, 12, 512, 64): (2, 1),
}


@pytest.mark.parametrize("case", list(S2D_CASES))
def test_conv_pipe_stride2_dgrad(ops, case, monkeypatch):
    """Data-gradient of a 3x3 / stride-2 / pad-1 conv on the pipelined kernel (four parity classes per unit) vs torch fp64, and
    against the merged-classes kernel it replaces (SST_CONV_PIPE=0)."""
    from srganst import _abi
    B, H, W, Cin, Cout = case
    tw, ks = S2D_CASES[case]
    L = _abi.lib()
    assert L.sst_conv_s2_dgrad_pipe_supported(B, H, W, Cin, Cout) == tw
    if tw:
        n_mt = (W // 2 // tw) * ((B * (H // 2) + 32 // tw - 1) // (32 // tw))
        assert L.sst_conv_s2_dgrad_pipe_ws_floats(B, H, W, Cin, Cout) == (ks * n_mt * (Cin // 32) * 4 * 1024 if ks > 1 else 0)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y = F.conv2d(x, w.double(), None, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    wp = ops.pack_conv_s2_dgrad(w.cuda())
    dyd = nhwc(dy).cuda()
    dx = ops.conv_s2_dgrad(dyd, wp, H, W, Cin)
    assert rel_err(nchw(dx.cpu()), x.grad) < TOL
    assert torch.equal(dx, ops.conv_s2_dgrad(dyd, wp, H, W, Cin))          # fixed-order reductions: bit-identical relaunch
    # the same launch with the BatchNorm / LeakyReLU backward partials of dx against the saved output of the layer below in its
    # epilogue: dx bit-identical, the column sums of the partials equal to the separate reduce pass over (dx, y)
    monkeypatch.setattr(ops, "S2D_EPI", True)
    yprev = torch.randn(B, H, W, Cin, generator=g).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.3).cuda()
    for kw in (dict(scale=sc, shift=sh), dict(scale=None, shift=None)):
        dx2, part = ops.conv_s2_dgrad(dyd, wp, H, W, Cin, epi=dict(y=yprev, slope_const=0.2, act=1, **kw))
        assert torch.equal(dx2, dx)
        if tw:
            assert part is not None and part.shape[0] == L.sst_conv_s2_dgrad_pipe_stat_tiles(B, H, W, Cin, Cout)
            ref = ops.bwd_reduce(dx, yprev, scale=kw["scale"], shift=kw["shift"], slope_const=0.2, act=1).double().sum(0)
            assert rel_err(part.double().sum(0).cpu(), ref.cpu()) < 1e-5
        else:
            assert part is None
    monkeypatch.setenv("SST_CONV_PIPE", "0")
    dx0 = ops.conv_s2_dgrad(dyd, wp, H, W, Cin)
    assert rel_err(dx.cpu(), dx0.cpu()) < 1e-5

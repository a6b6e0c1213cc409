"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

Activations are NHWC fp32 here; weights arrive in the reference layout [Cout,Cin,k,k] and are
packed by ``pack_conv``.  Used by the model graphs (gen_graph.py / disc_graph.py) and by the
per-kernel parity tests.  No arithmetic happens in Python.
"""
from __future__ import annotations

import os

import torch

from . import _abi
from ._abi import check, ptr, stream_ptr

OUT_NHWC, OUT_SHUFFLE, OUT_NCHW_CLAMP, OUT_UNSHUFFLE = 0, 1, 2, 3
ACT_NONE, ACT_SLOPE = 0, 1
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


# --- optional per-launch timing of the MFMA kernels (bench.py roofline leg): list of (kernel, flops, ev0, ev1)
PROFILE = None


# --- optional launch trace (bench.py roofline leg): kernel name -> list of (relaunch closure, flops).  The closures
# hold their tensors alive and re-issue exactly the same launch, so a kernel family can be replayed back-to-back
# from a hipGraph (GPU-bound timing, no host gaps).
TRACE = None


def _trace(name, flops, fn, *keep):
    if TRACE is not None:
        TRACE.setdefault(name, []).append((fn, flops, keep))


# --- the same for the HBM-bound kernels (structure-tensor loss, classifier GEMMs, Adam): name -> [(relaunch closure, ALGORITHMIC
# bytes of the launch, keep-alive)]
TRACE_HBM = None


def _trace_hbm(name, nbytes, fn, *keep):
    if TRACE_HBM is not None:
        TRACE_HBM.setdefault(name, []).append((fn, float(nbytes), keep))


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, name, flops):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((name, flops, e0, e1))


def _conv_name(B, H, W, cin, cout, ksize, stride, out_mode=0, fused_in=0):
    return _abi.lib().sst_conv_kernel_name(B, H, W, cin, cout, ksize, stride, int(out_mode) & 0xff, int(fused_in)).decode()


def _wgrad_name(B, H, W, cin, cout, ksize, stride, njobs):
    return _abi.lib().sst_conv_wgrad_kernel_name(B, H, W, cin, cout, ksize, stride, njobs).decode()


def _f32(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def pack_conv(w: torch.Tensor, mode: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    cout, cin, k, _ = w.shape
    o, i = (cin, cout) if mode else (cout, cin)
    n = _abi.lib().sst_conv_packed_floats(o, i, k)
    if out is None:
        out = _f32(n, like=w)
    assert out.numel() == n
    check(_abi.lib().sst_conv_pack(ptr(w), ptr(out), cout, cin, k, mode, stream_ptr()), "sst_conv_pack")
    return out


def conv_out_hw(h, w, k, stride):
    p = k // 2
    return (h + 2 * p - k) // stride + 1, (w + 2 * p - k) // stride + 1


def conv_fwd(x, wp, cout, ksize=3, stride=1, bias=None, in_scale=None, in_shift=None, in_slope=None,
             in_slope_const=0.0, in_act=ACT_NONE, residual=None, want_stats=False, out_mode=OUT_NHWC,
             want_pre=False, grp=0, out=None):
    """x [B,H,W,Cin] NHWC -> y (layout per out_mode); returns (y, y_pre|None, stats|None, cnt|None).
    grp > 0: the batch is B / grp passes of grp images each, in_scale / in_shift are [B / grp, Cin] (pipelined kernel only)."""
    B, H, W, cin = x.shape
    if grp and grp < B and in_scale is not None and not (out_mode == OUT_NHWC and residual is None and not want_pre
                                                         and conv_pipe_groups_ok(B, H, W, cin, cout, ksize, stride, grp)):
        raise _abi.HipPathError(f"conv_fwd: coefficient groups of {grp} images are not supported for this shape")
    ho, wo = conv_out_hw(H, W, ksize, stride)
    if out is not None:
        assert out_mode == OUT_NHWC and tuple(out.shape) == (B, ho, wo, cout) and out.is_contiguous()
        y = out
    elif out_mode == OUT_NHWC:
        y = _f32(B, ho, wo, cout, like=x)
    elif out_mode == OUT_SHUFFLE:
        y = _f32(B, 2 * ho, 2 * wo, cout // 4, like=x)
    elif out_mode == OUT_NCHW_CLAMP:
        y = _f32(B, cout, ho, wo, like=x)
    else:
        y = _f32(B, ho // 2, wo // 2, cout * 4, like=x)
    if out_mode == OUT_NHWC and residual is None and not want_pre and conv_pipe_tw(B, H, W, cin, cout, ksize, stride):
        y, stats, cnt, _ = _conv_pipe(x, wp, y, cout, ksize, stride, bias, in_scale, in_shift, in_slope, in_slope_const, in_act,
                                      want_stats, grp=grp)
        return y, None, stats, cnt
    if (out_mode == OUT_SHUFFLE and residual is None and not want_pre and not want_stats and CONV_NS
            and _abi.lib().sst_conv_ns_supported(B, H, W, cin, cout, ksize, stride, OUT_SHUFFLE)):
        # the up-sampling blocks' convs (model.py:157-161): N-split kernel with the PixelShuffle store
        y, _, _, _ = _conv_pipe(x, wp, y, cout, ksize, stride, bias, in_scale, in_shift, in_slope, in_slope_const, in_act, False,
                                out_mode=OUT_SHUFFLE)
        return y, None, None, None
    y_pre = torch.empty_like(y) if want_pre else None
    stats = cnt = None
    if want_stats:
        mt = _abi.lib().sst_conv_stat_tiles(B, H, W, cin, cout, ksize, stride)
        stats = _f32(mt, 2, cout, like=x)
        cnt = _f32(mt, like=x)
    e0 = _prof_begin()
    args = (ptr(x), ptr(wp), ptr(y), ptr(y_pre), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const),
            int(in_act), ptr(residual), ptr(stats), ptr(cnt), int(out_mode), B, H, W, cin, cout, ksize, stride)
    check(_abi.lib().sst_conv_fwd(*args, stream_ptr()), "sst_conv_fwd")
    flops = 2.0 * B * ho * wo * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, stride, out_mode) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_fwd(*args, stream_ptr()),
           x, wp, y, y_pre, bias, in_scale, in_shift, in_slope, residual, stats, cnt)
    return y, y_pre, stats, cnt


def conv_pipe_tw(B, H, W, cin, cout, ksize, stride):
    """Tile width of the persistent pipelined conv kernel (csrc/conv_pipe.hip) when it takes this NHWC-store shape, else 0.
    SST_CONV_PIPE=0 (dev switch) keeps every shape on the general kernel."""
    if os.environ.get("SST_CONV_PIPE", "1") == "0":
        return 0
    return _abi.lib().sst_conv_pipe_supported(B, H, W, cin, cout, ksize, stride)


def conv_pipe_groups_ok(B, H, W, cin, cout, ksize, stride, grp):
    """The pipelined conv kernel takes this shape AND passes of `grp` images end on its tile boundaries."""
    return bool(conv_pipe_tw(B, H, W, cin, cout, ksize, stride)) and \
        bool(_abi.lib().sst_conv_pipe_groups_ok(B, H, W, cin, cout, ksize, stride, int(grp)))


CONV_NS = os.environ.get("SST_CONV_NS", "1") != "0"      # the N-split form of the pipelined conv where it applies (0: dev A/B)


def _conv_pipe(x, wp, y, cout, ksize, stride, bias=None, in_scale=None, in_shift=None, in_slope=None, in_slope_const=0.0,
               in_act=ACT_NONE, want_stats=False, epi=None, grp=0, out_mode=OUT_NHWC):
    """sst_conv_pipe_fwd: forward statistics (want_stats) or backward partials (epi = dict(y, scale, shift, slope, slope_const,
    act)); returns (y, stats, cnt, partial)."""
    B, H, W, cin = x.shape
    L = _abi.lib()
    shp = (B, H, W, cin, cout, ksize, stride)
    stats = cnt = partial = None
    if want_stats:
        mt = L.sst_conv_pipe_stat_tiles(*shp)
        stats, cnt = _f32(mt, 2, cout, like=x), _f32(mt, like=x)
    e = epi or {}
    if epi is not None:
        partial = _f32(L.sst_conv_pipe_stat_tiles(*shp), 3, cout, like=x)
    if CONV_NS and L.sst_conv_ns_supported(*shp, int(out_mode)):
        # the N-split form (csrc/conv_nsplit.hip): same tiling, statistics rows and packed weights, no split-K workspace
        nargs = (ptr(x), ptr(wp), ptr(y), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const), int(in_act),
                 ptr(stats), ptr(cnt), ptr(e.get("y")), ptr(e.get("scale")), ptr(e.get("shift")), ptr(e.get("slope")),
                 float(e.get("slope_const", 0.0)), int(e.get("act", 0)), ptr(partial), *shp, int(out_mode), int(grp))
        e0 = _prof_begin()
        check(L.sst_conv_ns_fwd(*nargs, stream_ptr()), "sst_conv_ns_fwd")
        ho, wo = conv_out_hw(H, W, ksize, stride)
        flops = 2.0 * B * ho * wo * cout * cin * ksize * ksize
        name = f"conv_ns_kernel<{stride}, {L.sst_conv_ns_supported(*shp, int(out_mode))}>" if (PROFILE is not None or TRACE is not None) else ""
        _prof_end(e0, name, flops)
        _trace(name, flops, lambda: L.sst_conv_ns_fwd(*nargs, stream_ptr()),
               x, wp, y, bias, in_scale, in_shift, in_slope, stats, cnt, partial, *[v for v in e.values() if torch.is_tensor(v)])
        return y, stats, cnt, partial
    assert out_mode == OUT_NHWC
    nws = L.sst_conv_pipe_ws_floats(*shp)
    ws = _f32(nws, like=x) if nws else None
    args = (ptr(x), ptr(wp), ptr(y), ptr(bias), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const), int(in_act),
            ptr(stats), ptr(cnt), ptr(e.get("y")), ptr(e.get("scale")), ptr(e.get("shift")), ptr(e.get("slope")),
            float(e.get("slope_const", 0.0)), int(e.get("act", 0)), ptr(partial), ptr(ws), *shp, int(grp))
    e0 = _prof_begin()
    check(L.sst_conv_pipe_fwd_grp(*args, stream_ptr()), "sst_conv_pipe_fwd")
    ho, wo = conv_out_hw(H, W, ksize, stride)
    flops = 2.0 * B * ho * wo * cout * cin * ksize * ksize
    name = f"conv_pipe_kernel<{stride}, {L.sst_conv_pipe_supported(*shp)}, 0>" if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: L.sst_conv_pipe_fwd_grp(*args, stream_ptr()),
           x, wp, y, bias, in_scale, in_shift, in_slope, stats, cnt, partial, ws, *[v for v in e.values() if torch.is_tensor(v)])
    return y, stats, cnt, partial


_ONES = {}


def conv_fwd_resin(x, y2, bn_scale, bn_shift, wp, cout, ksize=3, bias=None, want_stats=False):
    """y = conv(h) with h = x + y2*bn_scale + bn_shift formed while staging (and returned): (y, h, stats|None, cnt|None)."""
    B, H, W, cin = x.shape
    key = (x.device, cin)
    ones = _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones(cin, device=x.device, dtype=torch.float32)
    y = _f32(B, H, W, cout, like=x)
    h = torch.empty_like(x)
    stats = cnt = None
    if want_stats:
        mt = _abi.lib().sst_conv_stat_tiles(B, H, W, cin, cout, ksize, 1)
        stats = _f32(mt, 2, cout, like=x)
        cnt = _f32(mt, like=x)
    args = (ptr(x), ptr(y2), ptr(ones), ptr(bn_scale), ptr(bn_shift), ptr(h), ptr(wp), ptr(y), ptr(bias), ptr(stats), ptr(cnt),
            B, H, W, cin, cout, ksize)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_fwd_resin(*args, stream_ptr()), "sst_conv_fwd_resin")
    flops = 2.0 * B * H * W * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, 1, 0, 1) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_fwd_resin(*args, stream_ptr()),
           x, y2, ones, bn_scale, bn_shift, h, wp, y, bias, stats, cnt)
    return y, h, stats, cnt


ACC_NREP = int(os.environ.get("SST_ACC_NREP", "16"))     # replicas of the fp64 BatchNorm accumulators (images are spread over them)


def conv_acc_supported(B, H, W, cin, cout, ksize=3, stride=1):
    return bool(_abi.lib().sst_conv_acc_supported(B, H, W, cin, cout, ksize, stride)) and os.environ.get("SST_ATOMIC_STATS", "1") != "0"


def conv_fwd_acc(x, wp, cout, ksize=3, in2=None, bias=None, in_slope=None, in_slope_const=0.0, in_act=ACT_NONE, in_acc=None,
                 in_bn=None, n=0.0, out_stats=None, run_stats=None, st_acc=None):
    """Forward conv in accumulator mode (see include/srganst.h: sst_conv_fwd_acc).
    in_acc: fp64 accumulators of the input's BatchNorm ([ACC_NREP,64,2]) + in_bn = (gamma, beta); out_stats = (mean, rstd, scale,
    shift) tensors to fill; run_stats = (running_mean, running_var) or None; st_acc: accumulators for this conv's output.
    in2: residual form (staged = x + in2*scale + shift).  Returns (y, h | None)."""
    B, H, W, cin = x.shape
    y = _f32(B, H, W, cout, like=x)
    h = torch.empty_like(x) if in2 is not None else None
    ones = None
    if in2 is not None:
        key = (x.device, cin)
        ones = _ONES.get(key)
        if ones is None:
            ones = _ONES[key] = torch.ones(cin, device=x.device, dtype=torch.float32)
    g, bta = in_bn if in_bn is not None else (None, None)
    om, orr, osc, osh = out_stats if out_stats is not None else (None, None, None, None)
    rm, rv = run_stats if run_stats is not None else (None, None)
    args = (ptr(x), ptr(in2), ptr(h), ptr(ones), ptr(wp), ptr(y), ptr(bias), ptr(in_slope), float(in_slope_const), int(in_act),
            _dptr(in_acc), ptr(g), ptr(bta), float(n), float(BN_EPS), float(BN_MOMENTUM), ptr(om), ptr(orr), ptr(osc), ptr(osh),
            ptr(rm), ptr(rv), _dptr(st_acc), ACC_NREP, B, H, W, cin, cout, ksize)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_fwd_acc(*args, stream_ptr()), "sst_conv_fwd_acc")
    flops = 2.0 * B * H * W * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, 1, 0, 1) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_fwd_acc(*args, stream_ptr()),
           x, in2, h, ones, wp, y, bias, in_slope, in_acc, g, bta, om, orr, osc, osh, rm, rv, st_acc)
    return y, h


def conv_dgrad_fused_acc(g, wd, cout, ksize=3, y2=None, in_scale=None, in_shift=None, in_slope=None, in_slope_const=0.0, in_act=0,
                         residual=None, epi_y=None, epi_scale=None, epi_shift=None, epi_slope=None, epi_slope_const=0.0, epi_act=0,
                         bw_in_acc=None, bn=None, n=0.0, dgamma=None, dbeta=None, dslope=None, bw_st_acc=None):
    """One BatchNorm-backward stage in accumulator mode (include/srganst.h: sst_conv_dgrad_fused_acc).  bn = (mean, rstd, gamma)
    of the BatchNorm being differentiated; returns (out, dy | None)."""
    B, H, W, cin = g.shape
    out = _f32(B, H, W, cout, like=g)
    dy = torch.empty_like(g) if y2 is not None else None
    mean, rstd, gamma = bn if bn is not None else (None, None, None)
    args = (ptr(g), ptr(y2), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const), int(in_act), ptr(dy), ptr(wd), ptr(out),
            ptr(residual), ptr(epi_y), ptr(epi_scale), ptr(epi_shift), ptr(epi_slope), float(epi_slope_const), int(epi_act),
            _dptr(bw_in_acc), ptr(mean), ptr(rstd), ptr(gamma), float(n), ptr(dgamma), ptr(dbeta), ptr(dslope), _dptr(bw_st_acc),
            ACC_NREP, B, H, W, cin, cout, ksize)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_dgrad_fused_acc(*args, stream_ptr()), "sst_conv_dgrad_fused_acc")
    flops = 2.0 * B * H * W * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, 1, 0, 1) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_dgrad_fused_acc(*args, stream_ptr()),
           g, y2, in_scale, in_shift, in_slope, dy, wd, out, residual, epi_y, epi_scale, epi_shift, epi_slope, bw_in_acc, mean, rstd, gamma,
           dgamma, dbeta, dslope, bw_st_acc)
    return out, dy


def _dptr(t):
    """Device pointer of a contiguous fp64 tensor (accumulators); None -> NULL."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
    return t.data_ptr()


def bn_finalize_acc(acc, n, gamma, beta, run_mean=None, run_var=None):
    """bn_finalize from fp64 accumulators [ACC_NREP, C, 2] -> (mean, rstd, scale, shift)."""
    C = gamma.numel()
    mean, rstd, scale, shift = (_f32(C, like=gamma) for _ in range(4))
    check(_abi.lib().sst_bn_finalize_acc(_dptr(acc), acc.shape[0], C, float(n), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_var),
                                         ptr(mean), ptr(rstd), ptr(scale), ptr(shift), float(BN_EPS), float(BN_MOMENTUM), stream_ptr()),
          "sst_bn_finalize_acc")
    return mean, rstd, scale, shift


def conv_wgrad(x, dy, dw_out, ksize=3, stride=1, in_scale=None, in_shift=None, in_slope=None, in_slope_const=0.0,
               in_act=ACT_NONE, accumulate=False, grp=0, defer_reduce=None):
    """dw_out [Cout,Cin,k,k] (reference layout) (+)= wgrad.  x [B,H,W,Cin], dy [B,Ho,Wo,Cout] NHWC.
    grp > 0: B / grp passes of grp images, in_scale / in_shift [B / grp, Cin] (all-taps tile kernel only).
    defer_reduce (a list): where the launch ends in a slab reduce, the reduce is not launched but appended as a job for
    wgrad_reduce_flush (one launch for all the layers of a backward pass); dw_out is complete only after that flush."""
    B, H, W, cin = x.shape
    cout = dy.shape[-1]
    ho, wo = conv_out_hw(H, W, ksize, stride)
    assert tuple(dy.shape) == (B, ho, wo, cout) and tuple(dw_out.shape) == (cout, cin, ksize, ksize)
    nch = _abi.lib().sst_conv_wgrad_chunks2(B, H, W, cin, cout, ksize, stride, 1)
    slab = _f32(nch * ksize * ksize * cout * cin, like=x)
    e0 = _prof_begin()
    pend = 0
    if defer_reduce is not None and PROFILE is None and TRACE is None:
        pend = _abi.lib().sst_conv_wgrad_pending_reduce(B, H, W, cin, cout, ksize, stride, int(in_scale is not None), int(in_act))
        if pend:
            assert dw_out.is_contiguous()
            defer_reduce.append((slab, dw_out, pend, ksize * ksize, cout, cin, int(bool(accumulate))))
    args = (ptr(x), ptr(dy), ptr(slab), ptr(dw_out), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const),
            int(in_act), B, H, W, cin, cout, stride, ksize, int(bool(accumulate)) | (4 if pend else 0), int(grp))
    check(_abi.lib().sst_conv_wgrad_grp(*args, stream_ptr()), "sst_conv_wgrad")
    flops = 2.0 * B * ho * wo * cout * cin * ksize * ksize
    name = ""
    if PROFILE is not None or TRACE is not None:
        name = _wgrad_name(B, H, W, cin, cout, ksize, stride, 1)
        name += "+c3m_reduce_kernel" if name == "wgrad_k3c3_mfma_kernel" else "+wgrad_reduce_kernel"
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_wgrad_grp(*args, stream_ptr()),
           x, dy, slab, dw_out, in_scale, in_shift, in_slope)
    return dw_out


WGRAD_REDUCE_MULTI = os.environ.get("SST_WGRAD_REDUCE_MULTI", "1") != "0"    # slab reduces of a backward pass in one launch (0: dev A/B)


def wgrad_reduce_flush(jobs):
    """The slab reduces conv_wgrad(defer_reduce=jobs) left behind, in one launch per 24 jobs (sst_wgrad_reduce_multi); empties `jobs`."""
    import ctypes
    import struct
    while jobs:
        js, jobs[:] = jobs[:24], jobs[24:]
        buf = b"".join(struct.pack("<QQiiiiii", slab.data_ptr(), dw.data_ptr(), nch, kk, cout, cin, acc, 0)
                       for slab, dw, nch, kk, cout, cin, acc in js)
        check(_abi.lib().sst_wgrad_reduce_multi(ctypes.c_char_p(buf), len(js), stream_ptr()), "sst_wgrad_reduce_multi")


def bn_finalize(stats, cnt, gamma, beta, run_mean=None, run_var=None, eps=BN_EPS, momentum=BN_MOMENTUM, groups=1, out=None):
    """-> (mean, rstd, scale, shift); updates run_mean/run_var in place when given (train mode).
    groups > 1: the tiles are `groups` equal consecutive ranges (passes batched as one tall image): outputs [groups, C], one momentum
    step of the running statistics per group, in order."""
    C = gamma.numel()
    shape = (C,) if groups == 1 else (groups, C)
    if out is not None:          # (mean, rstd, scale, shift) rows to fill: one pass's row of a [passes, C] table (disc_graph's pass arena)
        mean, rstd, scale, shift = out
        assert all(tuple(t.shape) == shape and t.is_contiguous() for t in out)
    else:
        mean, rstd, scale, shift = (_f32(*shape, like=gamma) for _ in range(4))
    check(_abi.lib().sst_bn_finalize_grp(ptr(stats), ptr(cnt), stats.shape[0], C, int(groups), ptr(gamma), ptr(beta), ptr(run_mean),
                                         ptr(run_var), ptr(mean), ptr(rstd), ptr(scale), ptr(shift), eps, momentum,
                                         stream_ptr()), "sst_bn_finalize")
    return mean, rstd, scale, shift


def bn_eval_affine(gamma, beta, run_mean, run_var, eps=BN_EPS):
    C = gamma.numel()
    scale, shift = _f32(C, like=gamma), _f32(C, like=gamma)
    check(_abi.lib().sst_bn_eval_affine(ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_var), ptr(scale), ptr(shift), C, eps,
                                        stream_ptr()), "sst_bn_eval_affine")
    return scale, shift


def bn_residual(y, scale, shift, res, res_slope=None):
    C = y.shape[-1]
    out = torch.empty_like(y)
    check(_abi.lib().sst_bn_residual(ptr(y), ptr(scale), ptr(shift), ptr(res), ptr(res_slope), ptr(out), y.numel() // C, C,
                                     stream_ptr()), "sst_bn_residual")
    return out


def bwd_reduce(g, y, g2=None, scale=None, shift=None, slope=None, slope_const=0.0, act=0, groups=1):
    """groups > 1: y is `groups` passes stacked along the rows, scale / shift [groups, C]; partial [groups * nblk, 3, C]."""
    C = y.shape[-1]
    R = y.numel() // C // groups
    nblk = _abi.lib().sst_bwd_reduce_blocks(R, C)
    partial = _f32(groups * nblk, 3, C, like=y)
    check(_abi.lib().sst_bwd_reduce_grp(ptr(g), ptr(g2), ptr(y), ptr(scale), ptr(shift), ptr(slope), float(slope_const), int(act),
                                        ptr(partial), R, C, int(groups), stream_ptr()), "sst_bwd_reduce")
    return partial


def bwd_finalize(partial, n, mean=None, rstd=None, gamma=None, dgamma=None, dbeta=None, dslope=None, accumulate=False, groups=1):
    """BN mode (mean given): fills dgamma/dbeta, returns (cA,cB,cC).  Otherwise dbeta (= dbias) / dslope only.
    groups > 1: partial holds `groups` consecutive ranges of blocks, mean / rstd and the returned coefficients are [groups, C], n counts
    one pass; dgamma / dbeta sum over the passes."""
    nblk, _, C = partial.shape
    cA = cB = cC = None
    if groups > 1:
        assert dslope is None and nblk % groups == 0
        if mean is not None:
            cA, cB, cC = (_f32(groups, C, like=partial) for _ in range(3))
        check(_abi.lib().sst_bwd_finalize_grp(ptr(partial), nblk // groups, C, float(n), int(groups), ptr(mean), ptr(rstd), ptr(gamma),
                                              ptr(dgamma), ptr(dbeta), ptr(cA), ptr(cB), ptr(cC), int(accumulate), stream_ptr()),
              "sst_bwd_finalize_grp")
        return cA, cB, cC
    if mean is not None:
        cA, cB, cC = (_f32(C, like=partial) for _ in range(3))
    if dslope is not None and C > 64:        # wide layer: keep the finalize channel-parallel (last-arriver slope sum)
        scratch = _f32((C + 63) // 64, like=partial)
        check(_abi.lib().sst_bwd_finalize_wide(ptr(partial), nblk, C, float(n), ptr(mean), ptr(rstd), ptr(gamma), ptr(dgamma),
                                               ptr(dbeta), ptr(cA), ptr(cB), ptr(cC), ptr(dslope), int(accumulate), ptr(scratch),
                                               _next_counter(partial.device), stream_ptr()), "sst_bwd_finalize_wide")
        return cA, cB, cC
    check(_abi.lib().sst_bwd_finalize(ptr(partial), nblk, C, float(n), ptr(mean), ptr(rstd), ptr(gamma), ptr(dgamma), ptr(dbeta),
                                      ptr(cA), ptr(cB), ptr(cC), ptr(dslope), int(accumulate), stream_ptr()),
          "sst_bwd_finalize")
    return cA, cB, cC


def act_bwd(g, y, g2=None, slope=None, slope_const=0.0, dbias=None, dslope=None, accumulate=False, unshuffle=False):
    """Backward through a slope activation without BatchNorm, one pass + one finalize: returns dy = act'(y) * (g + g2)
    (in the pre-PixelShuffle layout [B,H/2,W/2,4C] when unshuffle), fills dbias (sum of dy per stored channel) and dslope."""
    C = y.shape[-1]
    R = y.numel() // C
    uh = uw = 0
    if unshuffle:
        B, uh, uw, _ = y.shape
        dy = _f32(B, uh // 2, uw // 2, 4 * C, like=y)
    else:
        dy = torch.empty_like(y)
    Cs = dy.shape[-1]
    nblk = _abi.lib().sst_act_bwd_partial_blocks(dy.numel() // Cs)
    partial = _f32(nblk, 3, Cs, like=y)
    check(_abi.lib().sst_act_bwd_partial(ptr(g), ptr(g2), ptr(y), ptr(slope), float(slope_const), ptr(dy), ptr(partial), R, C, uh, uw,
                                         stream_ptr()), "sst_act_bwd_partial")
    if dbias is not None or dslope is not None:
        bwd_finalize(partial, R, dbeta=dbias, dslope=dslope, accumulate=accumulate)
    return dy


def bwd_apply(g, y, g2=None, scale=None, shift=None, slope=None, slope_const=0.0, act=0, cA=None, cB=None, cC=None,
              unshuffle=False, groups=1):
    """dy = cA*gz + cB*y + cC (BN input grad) or gz (activation grad only).  unshuffle: y is [B,2h,2w,C] and dy
    is stored as the pre-PixelShuffle tensor [B,h,w,4C].  groups > 1: passes stacked along the rows, coefficients [groups, C]."""
    C = y.shape[-1]
    if groups > 1:
        assert not unshuffle
        dy = torch.empty_like(y)
        check(_abi.lib().sst_bwd_apply_grp(ptr(g), ptr(g2), ptr(y), ptr(scale), ptr(shift), ptr(slope), float(slope_const), int(act),
                                           ptr(cA), ptr(cB), ptr(cC), ptr(dy), y.numel() // C // groups, C, int(groups), stream_ptr()),
              "sst_bwd_apply_grp")
        return dy
    uh = uw = 0
    if unshuffle:
        B, uh, uw, _ = y.shape
        dy = _f32(B, uh // 2, uw // 2, 4 * C, like=y)
    else:
        dy = torch.empty_like(y)
    check(_abi.lib().sst_bwd_apply(ptr(g), ptr(g2), ptr(y), ptr(scale), ptr(shift), ptr(slope), float(slope_const), int(act),
                                   ptr(cA), ptr(cB), ptr(cC), ptr(dy), y.numel() // C, C, uh, uw, stream_ptr()),
          "sst_bwd_apply")
    return dy


def add(a, b):
    out = torch.empty_like(a)
    check(_abi.lib().sst_add(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), "sst_add")
    return out


def transpose(x, to_nchw: bool, out=None):
    """NHWC [B,H,W,C] -> NCHW [B,C,H,W] (to_nchw) or the inverse."""
    if to_nchw:
        B, H, W, C = x.shape
        shape = (B, C, H, W)
    else:
        B, C, H, W = x.shape
        shape = (B, H, W, C)
    if out is None:
        out = _f32(*shape, like=x)
    assert tuple(out.shape) == shape and out.is_contiguous()
    check(_abi.lib().sst_transpose(ptr(x), ptr(out), B, C, H, W, int(to_nchw), stream_ptr()), "sst_transpose")
    return out


def clamp_bwd(g, pre, dbias=None, accumulate=False):
    """g, pre NCHW [B,C<=4,H,W] -> masked gradient NHWC [B,H,W,C]; dbias (+)= its column sums."""
    B, C, H, W = g.shape
    out = _f32(B, H, W, C, like=g)
    partial = _f32(_abi.lib().sst_clamp_bwd_blocks(B, H, W) * C, like=g)
    check(_abi.lib().sst_clamp_bwd(ptr(g), ptr(pre), ptr(out), ptr(partial), ptr(dbias), int(accumulate), B, C, H, W,
                                   stream_ptr()), "sst_clamp_bwd")
    return out


def pixel_loss_fwd(x, gt, mode, ws):
    n = x.numel()
    key = (x.device, n)
    if ws.get("key") != key:
        ws["key"] = key
        ws["partials"] = _f32(_abi.lib().sst_pixel_loss_blocks(n), like=x)
        ws["counter"] = torch.zeros(1, device=x.device, dtype=torch.int32)
    loss = _f32((), like=x)
    check(_abi.lib().sst_pixel_loss_fwd(ptr(x), ptr(gt), ptr(loss), ptr(ws["partials"]), ptr(ws["counter"]), n, mode,
                                        stream_ptr()), "sst_pixel_loss_fwd")
    return loss


def pixel_loss_bwd(x, gt, mode, scale_dev=None, scale_host=1.0, out=None, accumulate=False):
    if out is None:
        out = torch.empty_like(x)
        accumulate = False
    check(_abi.lib().sst_pixel_loss_bwd(ptr(x), ptr(gt), ptr(out), ptr(scale_dev), float(scale_host), int(accumulate), x.numel(),
                                        mode, stream_ptr()), "sst_pixel_loss_bwd")
    return out


def bce_logits(logits, target, want_loss=True, want_grad=False, scale_dev=None, scale_host=1.0, grad_out=None):
    loss = _f32((), like=logits) if want_loss else None
    dl = (grad_out if grad_out is not None else torch.empty_like(logits)) if want_grad else None
    check(_abi.lib().sst_bce_logits(ptr(logits), float(target), ptr(loss), ptr(dl), ptr(scale_dev), float(scale_host),
                                    logits.numel(), stream_ptr()), "sst_bce_logits")
    return loss, dl


def pack_conv_s2_dgrad(w, out=None):
    cout, cin, k, _ = w.shape
    assert k == 3
    n = _abi.lib().sst_conv_s2_dgrad_packed_floats(cout, cin)
    wp = out if out is not None and out.numel() == n and out.device == w.device else _f32(n, like=w)
    check(_abi.lib().sst_conv_s2_dgrad_pack(ptr(w), ptr(wp), cout, cin, stream_ptr()), "sst_conv_s2_dgrad_pack")
    return wp


S2D_EPI = os.environ.get("SST_S2D_EPI", "0") != "0"
S2D_EPI_GRP = os.environ.get("SST_S2D_EPI_GRP", "0") != "0"      # ... for passes batched as one tall image (grp > 0)


def conv_s2_dgrad(dy, wp, H, W, cin, epi=None, grp=0):
    """dy [B,Ho,Wo,Cout] -> dx [B,H,W,Cin] for y = conv3x3(x, stride 2, pad 1).
    epi = dict(y, scale, shift, slope, slope_const, act): also the BatchNorm / activation backward partials of dx against epi["y"]
    ([tiles,3,cin], the layout bwd_finalize consumes) where the pipelined kernel takes the shape -> (dx, partial | None)."""
    B, ho, wo, cout = dy.shape
    dx = _f32(B, H, W, cin, like=dy)
    L = _abi.lib()
    if os.environ.get("SST_CONV_PIPE", "1") != "0" and L.sst_conv_s2_dgrad_pipe_supported(B, H, W, cin, cout):
        nws = L.sst_conv_s2_dgrad_pipe_ws_floats(B, H, W, cin, cout)
        ws = _f32(nws, like=dy) if nws else None
        partial = None
        # measured on the G + D + ST step: 5.34 ms with the partials in the epilogue against 5.30 ms with the separate reduce pass (four
        # scattered epilogues per unit cost more than the three reduce launches they replace) - off unless SST_S2D_EPI=1
        grouped = bool(grp) and grp < B
        if epi is not None and ((S2D_EPI and not grouped) or
                                (grouped and S2D_EPI_GRP and L.sst_conv_s2_dgrad_pipe_groups_ok(B, H, W, cin, cout, int(grp)))):
            partial = _f32(L.sst_conv_s2_dgrad_pipe_stat_tiles(B, H, W, cin, cout), 3, cin, like=dy)
        e = epi or {}
        args = (ptr(dy), ptr(wp), ptr(dx), ptr(ws), ptr(e.get("y")) if partial is not None else None,
                ptr(e.get("scale")) if partial is not None else None, ptr(e.get("shift")) if partial is not None else None,
                ptr(e.get("slope")) if partial is not None else None, float(e.get("slope_const", 0.0)), int(e.get("act", 0)),
                ptr(partial), B, H, W, cin, cout, int(grp) if partial is not None else 0)
        e0 = _prof_begin()
        check(L.sst_conv_s2_dgrad_pipe_bwdstats_grp(*args, stream_ptr()), "sst_conv_s2_dgrad_pipe")
        if PROFILE is not None or TRACE is not None:
            name, flops = f"conv_pipe_kernel<1, {L.sst_conv_s2_dgrad_pipe_supported(B, H, W, cin, cout)}, 1>", 2.0 * B * ho * wo * cout * cin * 9
            _prof_end(e0, name, flops)
            _trace(name, flops, lambda: L.sst_conv_s2_dgrad_pipe_bwdstats_grp(*args, stream_ptr()), dy, wp, dx, ws, partial,
                   e.get("y"), e.get("scale"), e.get("shift"), e.get("slope"))
        return (dx, partial) if epi is not None else dx
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_s2_dgrad(ptr(dy), ptr(wp), ptr(dx), B, H, W, cin, cout, stream_ptr()), "sst_conv_s2_dgrad")
    if PROFILE is not None or TRACE is not None:
        name, flops = _abi.lib().sst_conv_s2_dgrad_kernel_name(B, H, W, cin, cout, 0).decode(), 2.0 * B * ho * wo * cout * cin * 9
        _prof_end(e0, name, flops)
        _trace(name, flops, lambda: _abi.lib().sst_conv_s2_dgrad(ptr(dy), ptr(wp), ptr(dx), B, H, W, cin, cout, stream_ptr()), dy, wp, dx)
    return (dx, None) if epi is not None else dx


def conv_s2_dgrad_fused(g, y2, wp, H, W, cin, cA=None, cB=None, cC=None, in_scale=None, in_shift=None, in_slope=None,
                        in_slope_const=0.0, in_act=0, epi_y=None, epi_scale=None, epi_shift=None, epi_slope=None, epi_slope_const=0.0,
                        epi_act=0):
    """conv_dgrad_fused for a stride-2 conv: g / y2 [B,Ho,Wo,Cout] (conv output side), returns (dx [B,H,W,cin], dy, partial | None)."""
    B, ho, wo, cout = g.shape
    dx = _f32(B, H, W, cin, like=g)
    dy = torch.empty_like(g)
    partial = _f32(_abi.lib().sst_conv_s2_dgrad_tiles(B, H, W), 3, cin, like=g) if epi_y is not None else None
    args = (ptr(g), ptr(y2), ptr(cA), ptr(cB), ptr(cC), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const), int(in_act),
            ptr(dy), ptr(wp), ptr(dx), ptr(epi_y), ptr(epi_scale), ptr(epi_shift), ptr(epi_slope), float(epi_slope_const), int(epi_act),
            ptr(partial), B, H, W, cin, cout)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_s2_dgrad_fused(*args, stream_ptr()), "sst_conv_s2_dgrad_fused")
    flops = 2.0 * B * ho * wo * cout * cin * 9
    name = _abi.lib().sst_conv_s2_dgrad_kernel_name(B, H, W, cin, cout, 1).decode() if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_s2_dgrad_fused(*args, stream_ptr()),
           g, y2, cA, cB, cC, in_scale, in_shift, in_slope, dy, wp, dx, epi_y, epi_scale, epi_shift, epi_slope, partial)
    return dx, dy, partial


def linear_fwd(x, w, bias, out=None):
    M, K = x.shape
    N = w.shape[0]
    y = out if out is not None else _f32(M, N, like=x)
    assert tuple(y.shape) == (M, N) and y.is_contiguous()
    slab = _f32(_abi.lib().sst_linear_ksplit(M, N, K) * M * N, like=x)
    args = (ptr(x), ptr(w), ptr(bias), ptr(y), ptr(slab), M, N, K)
    check(_abi.lib().sst_linear_fwd(*args, stream_ptr()), "sst_linear_fwd")
    _trace_hbm("linear_fwd_kernel", 4.0 * (N * K + M * K + M * N), lambda: _abi.lib().sst_linear_fwd(*args, stream_ptr()), x, w, bias, y, slab)
    return y


def linear_dgrad(dy, w, nhwc=None):
    """dx [M,K]; nhwc=(C,HW): K indexes an NCHW flatten and dx is written as NHWC [M,HW,C]."""
    M, N = dy.shape
    K = w.shape[1]
    dx = _f32(M, K, like=dy)
    c, hw = nhwc if nhwc else (0, 0)
    args = (ptr(dy), ptr(w), ptr(dx), M, N, K, c, hw)
    check(_abi.lib().sst_linear_dgrad(*args, stream_ptr()), "sst_linear_dgrad")
    _trace_hbm("linear_dgrad_kernel", 4.0 * (N * K + M * K + M * N), lambda: _abi.lib().sst_linear_dgrad(*args, stream_ptr()), dy, w, dx)
    return dx


def linear_wgrad(dy, x, dw, db=None, accumulate=False):
    M, N = dy.shape
    K = x.shape[1]
    args = (ptr(dy), ptr(x), ptr(dw), ptr(db), M, N, K, int(accumulate))
    check(_abi.lib().sst_linear_wgrad(*args, stream_ptr()), "sst_linear_wgrad")
    _trace_hbm("linear_wgrad_kernel", 4.0 * (N * K * (2 if accumulate else 1) + M * K + M * N),
               lambda: _abi.lib().sst_linear_wgrad(*args, stream_ptr()), dy, x, dw, db)


def head_fwd(h, w, b, slope):
    M, K = h.shape
    N = w.shape[0]
    y = _f32(M, N, like=h)
    check(_abi.lib().sst_head_fwd(ptr(h), ptr(w), ptr(b), ptr(y), M, N, K, float(slope), stream_ptr()), "sst_head_fwd")
    return y


def head_bwd(h, w, dy, slope, dw=None, db=None, accumulate=False):
    M, K = h.shape
    N = w.shape[0]
    dh = torch.empty_like(h)
    check(_abi.lib().sst_head_bwd(ptr(h), ptr(w), ptr(dy), ptr(dh), ptr(dw), ptr(db), M, N, K, float(slope), int(accumulate),
                                  stream_ptr()), "sst_head_bwd")
    return dh


def flatten_act(y, scale, shift, slope, act=1, grp=0, out=None):
    """NHWC [B,H,W,C] -> [B, C*H*W] in NCHW-flatten order with act(y*scale+shift) applied (grp > 0: scale / shift [B / grp, C])."""
    B, H, W, C = y.shape
    flat = out if out is not None else _f32(B, C * H * W, like=y)
    assert tuple(flat.shape) == (B, C * H * W) and flat.is_contiguous()
    check(_abi.lib().sst_flatten_act_grp(ptr(y), ptr(scale), ptr(shift), float(slope), int(act), ptr(flat), B, H * W, C, int(grp),
                                         stream_ptr()), "sst_flatten_act")
    return flat


def wgrad_c3(big, small, dw_out, kind, in_slope=None, in_slope_const=0.0, in_act=ACT_NONE, accumulate=False):
    """Weight gradient of a 9x9 conv with a 3-channel side.  kind 0: conv3 (C->3): big = conv input, small = dY,
    dw_out [3,C,9,9].  kind 1: conv1 (3->C): big = dY, small = conv input, dw_out [C,3,9,9]."""
    B, H, W, C = big.shape
    assert tuple(small.shape) == (B, H, W, 3)
    assert tuple(dw_out.shape) == ((3, C, 9, 9) if kind == 0 else (C, 3, 9, 9))
    slab = _f32(_abi.lib().sst_wgrad_c3_slab_floats(B, H, W, C), like=big)
    e0 = _prof_begin()
    check(_abi.lib().sst_wgrad_c3(ptr(big), ptr(small), ptr(slab), ptr(dw_out), ptr(in_slope), float(in_slope_const), int(in_act),
                                  kind, B, H, W, C, int(accumulate), stream_ptr()), "sst_wgrad_c3")
    _prof_end(e0, "wgrad_c3_kernel+reduce", 2.0 * B * H * W * C * 3 * 81)
    return dw_out


def wgrad_c3_supported(C, ksize=9):
    return bool(_abi.lib().sst_wgrad_c3_supported(C, ksize))


# ------------------------------------------------------------------------------------------------
PACK_FWD, PACK_DGRAD, PACK_C3_FWD, PACK_C3_DGRAD, PACK_TO3 = 0, 1, 2, 3, 4      # PackJob modes (csrc/conv_fwd.hip)


class PackPlan:
    """All conv weights of a network packed by ONE kernel launch (sst_conv_pack_multi).  The job table and the
    packed buffers are persistent, so the launch is graph-capturable; rebuild when a parameter moves."""

    def __init__(self, weights, modes, extras=()):
        dev = weights[0].device
        self.ptrs = tuple(w.data_ptr() for w in weights) + self._extras_sig(extras)
        self.out, rows, blk = [], [], 0
        for w, m in zip(weights, modes):
            cout, cin, k, _ = w.shape
            if m == PACK_C3_FWD or m == PACK_C3_DGRAD:          # 9x9 with a 3-channel side (conv9_c3_fwd's layout)
                n = _abi.lib().sst_conv9_c3_packed_floats(cin if m == PACK_C3_DGRAD else cout)
            elif m == PACK_TO3:                                 # conv9_to3_fwd's layout
                n = _abi.lib().sst_conv9_to3_packed_floats(cin)
            else:
                o, i = (cin, cout) if m else (cout, cin)
                n = _abi.lib().sst_conv_packed_floats(o, i, k)
            assert n < 2 ** 31                                  # the pack kernel's index arithmetic is 32-bit
            wp = torch.empty(n, device=dev, dtype=torch.float32)
            self.out.append(wp)
            rows.append([w.data_ptr(), wp.data_ptr(), cout | (cin << 32), (k * k) | (int(m) << 32), n, blk])
            blk += (n + 1023) // 1024
        # extras ride along in the same launch (csrc/conv_fwd.hip: pack_multi_kernel modes 5 / 6): ("zero", tensor) clears it,
        # ("add", int64 tensor, k) adds k to every element - the statistics accumulators and BatchNorm batch counters of a forward
        for e in extras:
            if e[0] == "zero":
                t = e[1]
                assert t.is_contiguous() and t.element_size() % 4 == 0
                n = t.numel() * (t.element_size() // 4)
                rows.append([0, t.data_ptr(), 0, 5 << 32, n, blk])
            else:
                t, k = e[1], int(e[2])
                assert e[0] == "add" and t.dtype == torch.int64 and t.is_contiguous() and 0 <= k < 2 ** 31
                n = t.numel()
                rows.append([0, t.data_ptr(), k, 6 << 32, n, blk])
            blk += (n + 1023) // 1024
        self.njobs = len(rows)
        self.blocks = blk
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)

    @staticmethod
    def _extras_sig(extras):
        return tuple((e[0], e[1].data_ptr(), e[1].numel(), int(e[2]) if len(e) > 2 else 0) for e in extras)

    def matches(self, weights, extras=()):
        return self.ptrs == tuple(w.data_ptr() for w in weights) + self._extras_sig(extras)

    def run(self):
        check(_abi.lib().sst_conv_pack_multi(ptr(self.table), self.njobs, self.blocks, stream_ptr()), "sst_conv_pack_multi")
        return self.out


def packed_weights(cache: dict, key, weights, modes, extras=()):
    """cache: a dict living on the module.  Returns the list of packed tensors (freshly re-packed); `extras`: see PackPlan."""
    plan = cache.get(key)
    if plan is None or not plan.matches(weights, extras):
        plan = PackPlan(weights, modes, extras)
        cache[key] = plan
    return plan.run()


_COUNTERS = {}


def _next_counter(device):
    """One zeroed 32-bit word for a last-arriver kernel.  Kernels leave their word zero again, so a ring is enough;
    under graph capture every captured launch simply keeps its own word."""
    st = _COUNTERS.get(device)
    if st is None:
        st = [torch.zeros(8192, device=device, dtype=torch.int32), 0]
        _COUNTERS[device] = st
    i = st[1]
    st[1] = (i + 1) % 8192
    return st[0].data_ptr() + 4 * i


def flatten_bn_counters(module):
    """Make every BatchNorm's num_batches_tracked a view of one int64 tensor so a train-mode forward bumps them
    with a single add (33 launches -> 1 for the generator).  Idempotent; re-done if the buffers were moved."""
    bns = [m for m in module.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    if not bns:
        return None
    flat = getattr(module, "_nbt_flat", None)
    ok = flat is not None and flat.device == bns[0].num_batches_tracked.device and all(
        bn.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, bn in enumerate(bns))
    if not ok:
        flat = torch.stack([bn.num_batches_tracked.detach().to(torch.int64) for bn in bns])
        for i, bn in enumerate(bns):
            bn._buffers["num_batches_tracked"] = flat[i]
        module._nbt_flat = flat
    return flat


# ------------------------------------------------------------------------------------------------
# Second HIP stream for work that is off the critical chain of backward (weight gradients): under hipGraph
# capture the fork/join below becomes two parallel branches of the graph.
OVERLAP = False     # measured (round 1, MI355X): with the band kernels the side stream costs 7 % of the step - kept as an option
_SIDE = {}

# What hipStreamEndCapture (ROCm 7.2) accepts, established with tools/capture_probe.py (each pattern in its own process, trivial
# kernels): any number of streams forked from the capture's ORIGIN stream, forks of forks, events between sibling branches, RCCL
# collectives (their stream forks from the issuing branch) - as long as every SECOND-level stream (one that entered the capture through
# a stream other than the origin) is joined into the origin stream itself.  A dependency edge - wait_stream or wait_event alike -
# from a second-level stream into a first-level one makes hipStreamEndCapture fault (segmentation fault inside the runtime, no error
# code): that was the "third concurrent branch" crash of round 2 (the discriminator's two passes on two streams inside the side branch
# of the merged iteration, and side-stream weight gradients there, both joined the side branch).  The engines record the origin
# stream of an open capture here and every join of a helper stream goes through check_capture_join, which refuses that edge.
CAPTURE_ORIGIN = None
TOPOLOGY_ERROR = None       # message of the last refusal (the capture's own teardown may raise over the exception)


class CaptureTopologyError(RuntimeError):
    pass


def check_capture_join(dst_stream):
    """Call before dst_stream.wait_stream(child) / wait_event(event of a child stream) where `child` was forked from dst_stream: legal
    in eager mode and when dst_stream is the origin of the open capture, refused otherwise (see CAPTURE_ORIGIN)."""
    global TOPOLOGY_ERROR
    if CAPTURE_ORIGIN is not None and torch.cuda.is_current_stream_capturing() and dst_stream != CAPTURE_ORIGIN:
        TOPOLOGY_ERROR = (
            "a stream forked from a side branch of an open hipGraph capture may only be joined into the capture's origin stream: "
            "joining it into the side branch makes hipStreamEndCapture fault on ROCm 7.2 (ops.CAPTURE_ORIGIN, tools/capture_probe.py). "
            "Run this schedule outside the merged iteration (KERNEL.OVERLAP_GD = False) or without the helper stream.")
        raise CaptureTopologyError(TOPOLOGY_ERROR)


class SideStream:
    """with SideStream(tensors...):  the body runs on the side stream after everything issued so far on the
    current stream; `tensors` (inputs/outputs touched by the body) are recorded for the caching allocator."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if not OVERLAP:
            return self
        main = torch.cuda.current_stream()
        side = _SIDE.get(main.device)
        if side is None:
            side = torch.cuda.Stream(device=main.device)
            _SIDE[main.device] = side
        side.wait_stream(main)
        for t in self.tensors:
            t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if OVERLAP:
            self.ctx.__exit__(*exc)
        return False


def join_side():
    """The current stream waits for the side stream (call once before the gradients are consumed)."""
    if OVERLAP:
        main = torch.cuda.current_stream()
        side = _SIDE.get(main.device)
        if side is not None:
            check_capture_join(main)
            main.wait_stream(side)


def conv9_c3_fwd(x3, w, mode, bias=None, wp=None):
    """9x9 conv from a 3-channel NHWC tensor.  mode 0: w [Cout,3,9,9] (conv1 forward).  mode 1: w [3,C,9,9] and the
    result is the data-gradient of that conv (C output channels).  wp: w already packed (PackPlan mode PACK_C3_*)."""
    B, H, W, _ = x3.shape
    cout = w.shape[1] if mode else w.shape[0]
    if wp is None:
        wp = _f32(_abi.lib().sst_conv9_c3_packed_floats(cout), like=w)
        check(_abi.lib().sst_conv9_c3_pack(ptr(w), ptr(wp), w.shape[0], w.shape[1], mode, stream_ptr()), "sst_conv9_c3_pack")
    y = _f32(B, H, W, cout, like=x3)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv9_c3_fwd(ptr(x3), ptr(wp), ptr(y), ptr(bias), B, H, W, cout, stream_ptr()), "sst_conv9_c3_fwd")
    _prof_end(e0, "conv9_c3_fwd_kernel", 2.0 * B * H * W * cout * 3 * 81)
    return y


def conv9_to3_fwd(x, w, bias=None, in_slope=None, in_slope_const=0.0, in_act=ACT_NONE, want_pre=False, wp=None):
    """x [B,H,W,C] NHWC -> (clamp(conv9x9(act(x)) + bias, 0, 1) as NCHW [B,3,H,W], pre-clamp copy or None).
    wp: w already packed (PackPlan mode PACK_TO3)."""
    B, H, W, C = x.shape
    assert tuple(w.shape) == (3, C, 9, 9)
    if wp is None:
        wp = _f32(_abi.lib().sst_conv9_to3_packed_floats(C), like=w)
        check(_abi.lib().sst_conv9_to3_pack(ptr(w), ptr(wp), C, stream_ptr()), "sst_conv9_to3_pack")
    y = _f32(B, 3, H, W, like=x)
    y_pre = torch.empty_like(y) if want_pre else None
    e0 = _prof_begin()
    check(_abi.lib().sst_conv9_to3_fwd(ptr(x), ptr(wp), ptr(y), ptr(y_pre), ptr(bias), ptr(in_slope), float(in_slope_const),
                                       int(in_act), B, H, W, C, stream_ptr()), "sst_conv9_to3_fwd")
    _prof_end(e0, "conv9_to3_fwd_kernel", 2.0 * B * H * W * C * 3 * 81)
    return y, y_pre


def bwd_reduce_apply(g, y, n, g2=None, scale=None, shift=None, slope=None, slope_const=0.0, act=0, mean=None, rstd=None,
                     gamma=None, dgamma=None, dbeta=None, dslope=None, accumulate=False, unshuffle=False, groups=1):
    """Backward through [BatchNorm ->] activation: row-parallel reduction, channel-parallel finalize (BN-backward
    coefficients + dgamma / dbeta / dslope), elementwise apply.  Returns dy (pre-PixelShuffle layout when unshuffle).
    groups > 1: passes stacked along the rows (n = elements per channel of ONE pass)."""
    part = bwd_reduce(g, y, g2=g2, scale=scale, shift=shift, slope=slope, slope_const=slope_const, act=act, groups=groups)
    cA, cB, cC = bwd_finalize(part, n, mean, rstd, gamma, dgamma, dbeta, dslope, accumulate, groups=groups)
    return bwd_apply(g, y, g2=g2, scale=scale, shift=shift, slope=slope, slope_const=slope_const, act=act, cA=cA, cB=cB, cC=cC,
                     unshuffle=unshuffle, groups=groups)


def transpose_affine(x, to_nchw: bool, scale, shift=None):
    if to_nchw:
        B, H, W, C = x.shape
        out = _f32(B, C, H, W, like=x)
    else:
        B, C, H, W = x.shape
        out = _f32(B, H, W, C, like=x)
    check(_abi.lib().sst_transpose_affine(ptr(x), ptr(out), B, C, H, W, int(to_nchw), ptr(scale), ptr(shift), stream_ptr()),
          "sst_transpose_affine")
    return out


def maxpool_relu_fwd(y):
    B, H, W, C = y.shape
    out = _f32(B, H // 2, W // 2, C, like=y)
    check(_abi.lib().sst_maxpool_relu_fwd(ptr(y), ptr(out), B, H, W, C, stream_ptr()), "sst_maxpool_relu_fwd")
    return out


def maxpool_relu_bwd(g, y):
    B, H, W, C = y.shape
    dy = torch.empty_like(y)
    check(_abi.lib().sst_maxpool_relu_bwd(ptr(g), ptr(y), ptr(dy), B, H, W, C, stream_ptr()), "sst_maxpool_relu_bwd")
    return dy


def conv_dgrad_bwdstats(dy, wd, cout, ksize, epi_y, residual=None, epi_scale=None, epi_shift=None, epi_slope=None,
                        epi_slope_const=0.0, epi_act=0, grp=0):
    """Stride-1 data-gradient g = conv(dy, wd) (+ residual) that also returns the BN/activation backward partials of g
    against epi_y ([mtiles,3,cout], the layout bwd_finalize consumes)."""
    B, H, W, cin = dy.shape
    g = _f32(B, H, W, cout, like=dy)
    if residual is None and conv_pipe_tw(B, H, W, cin, cout, ksize, 1):
        _, _, _, partial = _conv_pipe(dy, wd, g, cout, ksize, 1, epi=dict(y=epi_y, scale=epi_scale, shift=epi_shift, slope=epi_slope,
                                                                         slope_const=epi_slope_const, act=epi_act), grp=grp)
        return g, partial
    if grp and grp < B and epi_scale is not None:
        raise _abi.HipPathError("conv_dgrad_bwdstats: coefficient groups need the pipelined kernel")
    mt = _abi.lib().sst_conv_stat_tiles(B, H, W, cin, cout, ksize, 1)
    partial = _f32(mt, 3, cout, like=dy)
    args = (ptr(dy), ptr(wd), ptr(g), ptr(residual), ptr(epi_y), ptr(epi_scale), ptr(epi_shift), ptr(epi_slope),
            float(epi_slope_const), int(epi_act), ptr(partial), B, H, W, cin, cout, ksize)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_dgrad_bwdstats(*args, stream_ptr()), "sst_conv_dgrad_bwdstats")
    flops = 2.0 * B * H * W * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, 1) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_dgrad_bwdstats(*args, stream_ptr()),
           dy, wd, g, residual, epi_y, epi_scale, epi_shift, epi_slope, partial)
    return g, partial


def bwd_finalize_apply(part, g, y, n, g2=None, scale=None, shift=None, slope=None, slope_const=0.0, act=0, mean=None, rstd=None,
                       gamma=None, dgamma=None, dbeta=None, dslope=None, accumulate=False, unshuffle=False, groups=1):
    """bwd_reduce_apply with the reduction already done (partials from conv_dgrad_bwdstats)."""
    cA, cB, cC = bwd_finalize(part, n, mean, rstd, gamma, dgamma, dbeta, dslope, accumulate, groups=groups)
    return bwd_apply(g, y, g2=g2, scale=scale, shift=shift, slope=slope, slope_const=slope_const, act=act, cA=cA, cB=cB, cC=cC,
                     unshuffle=unshuffle, groups=groups)


def conv_dgrad_fused(g, y2, wd, cout, ksize, cA=None, cB=None, cC=None, in_scale=None, in_shift=None, in_slope=None,
                     in_slope_const=0.0, in_act=0, residual=None, epi_y=None, epi_scale=None, epi_shift=None, epi_slope=None,
                     epi_slope_const=0.0, epi_act=0):
    """One BatchNorm-backward stage in one launch: dy = BN/activation backward of g against y2 (computed while staging,
    also returned for the weight gradient), out = stride-1 data-gradient conv(dy) (+ residual), optional backward
    partials of `out` against epi_y.  Returns (out, dy, partial | None)."""
    B, H, W, cin = g.shape
    out = _f32(B, H, W, cout, like=g)
    dy = torch.empty_like(g)
    partial = None
    if epi_y is not None:
        partial = _f32(_abi.lib().sst_conv_stat_tiles(B, H, W, cin, cout, ksize, 1), 3, cout, like=g)
    args = (ptr(g), ptr(y2), ptr(cA), ptr(cB), ptr(cC), ptr(in_scale), ptr(in_shift), ptr(in_slope), float(in_slope_const),
            int(in_act), ptr(dy), ptr(wd), ptr(out), ptr(residual), ptr(epi_y), ptr(epi_scale), ptr(epi_shift), ptr(epi_slope),
            float(epi_slope_const), int(epi_act), ptr(partial), B, H, W, cin, cout, ksize)
    e0 = _prof_begin()
    check(_abi.lib().sst_conv_dgrad_fused(*args, stream_ptr()), "sst_conv_dgrad_fused")
    flops = 2.0 * B * H * W * cout * cin * ksize * ksize
    name = _conv_name(B, H, W, cin, cout, ksize, 1, 0, 1) if (PROFILE is not None or TRACE is not None) else ""
    _prof_end(e0, name, flops)
    _trace(name, flops, lambda: _abi.lib().sst_conv_dgrad_fused(*args, stream_ptr()),
           g, y2, cA, cB, cC, in_scale, in_shift, in_slope, dy, wd, out, residual, epi_y, epi_scale, epi_shift, epi_slope, partial)
    return out, dy, partial


def flat_layout(params):
    """Offsets (in floats) of the per-parameter views inside a flat buffer, and its total size.  Every view starts on a
    64-byte boundary (vector loads in the flat Adam kernel, aligned RCCL message); pad words are don't-care."""
    offs, off = [], 0
    for t in params:
        offs.append(off)
        off += (t.numel() + 15) // 16 * 16
    return offs, off


def flat_grads(module, names, params):
    """One flat fp32 buffer + per-parameter views (reference order).  The buffer is remembered on the module so that
    the data-parallel exchange can all-reduce it in place as ONE message (no flatten / unflatten copies) and the flat
    Adam (srganst.optim.FlatAdam) can consume it as one array.
    The buffers are PERSISTENT: a ring of FLAT_RING zero-initialised buffers per module, allocated at the first call (always an eager
    warm-up call: never inside a graph capture) and handed out in turn.  The pad words between the views (flat_layout) are written by
    nobody, so they stay zero for good - they travel through the flat Adam and the all-reduce with the real gradients, and an
    uninitialised NaN there would poison any norm / isfinite check over the flat buffer.  A buffer comes round again after
    FLAT_RING - 1 other backward passes of the module (at most two per step are alive at once: the two-stream discriminator step)."""
    offs, total = flat_layout(params)
    dev = params[0].device
    ring = module.__dict__.get("_flat_ring")
    if ring is None or ring["total"] != total or ring["device"] != dev:
        if torch.cuda.is_current_stream_capturing():
            raise _abi.HipPathError("flat_grads: the gradient buffers must exist before a graph capture (run an eager warm-up step first)")
        ring = {"total": total, "device": dev, "next": 0,
                "bufs": [torch.zeros(total, device=dev, dtype=torch.float32) for _ in range(FLAT_RING)]}
        module.__dict__["_flat_ring"] = ring
    flat = ring["bufs"][ring["next"]]
    ring["next"] = (ring["next"] + 1) % FLAT_RING
    views = {n: flat[o:o + t.numel()].view(t.shape) for n, t, o in zip(names, params, offs)}
    # remember the order of use: with two backward passes per step (D on gt and on sr) autograd accumulates into the
    # FIRST pass's buffer, which is then the one holding p.grad
    lst = module.__dict__.setdefault("_flat_grads", [])
    lst[:] = [t for t in lst if t is not flat] + [flat]
    del lst[:-4]
    return views


FLAT_RING = 4


def flatten_params(module):
    """Move every parameter of `module` into ONE flat fp32 buffer (flat_layout order = named_parameters order, the same
    layout flat_grads gives the gradients) and re-point the nn.Parameters at views of it.  Values, names, shapes and
    state_dict are unchanged.  Returns (flat, offsets, params)."""
    params = [p for _, p in module.named_parameters()]
    ent = module.__dict__.get("_flat_params")
    if ent is not None and all(p.data_ptr() == ent[0].data_ptr() + 4 * o for p, o in zip(params, ent[1])):
        return ent[0], ent[1], params
    offs, total = flat_layout(params)
    flat = torch.zeros(total, device=params[0].device, dtype=torch.float32)
    with torch.no_grad():
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
    module.__dict__["_flat_params"] = (flat, offs)
    return flat, offs, params


def adam_flat(p, g, m, v, lr_dev, steps, beta1, beta2, eps, weight_decay):
    assert p.numel() == g.numel() == m.numel() == v.numel() and p.numel() % 4 == 0
    args = (ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(lr_dev), ptr(steps), steps.numel(), float(beta1), float(beta2), float(eps),
            float(weight_decay))
    check(_abi.lib().sst_adam_flat(*args, stream_ptr()), "sst_adam_flat")
    _trace_hbm("adam_flat_kernel", 28.0 * p.numel(), lambda: _abi.lib().sst_adam_flat(*args, stream_ptr()), p, g, m, v, lr_dev, steps)


class WgradGroup:
    """Collects weight-gradient jobs of identical shape and issues them as ONE launch (+ one grouped slab reduce).
    The slab is cached per (shape, count); the job table (device pointers of every layer) is a kernel argument."""
    _cache = {}

    def __init__(self):
        self.jobs = []

    def add(self, x, dy, dw_out, ksize=3, stride=1, in_scale=None, in_shift=None, in_slope=None, in_slope_const=0.0, in_act=ACT_NONE):
        self.jobs.append((x, dy, dw_out, ksize, stride, in_scale, in_shift, in_slope, float(in_slope_const), int(in_act)))

    def run(self):
        import struct
        if not self.jobs:
            return
        groups = {}
        for j in self.jobs:
            x, dy, dw, k, s = j[:5]
            groups.setdefault((tuple(x.shape), tuple(dy.shape), k, s), []).append(j)
        MAXJ = 40                          # jobs per launch: the table is a kernel argument (csrc/conv_wgrad.hip: WG_TAB_MAX)
        work = [((xs, dys, k, s), js[i:i + MAXJ]) for (xs, dys, k, s), js in groups.items() for i in range(0, len(js), MAXJ)]
        for (xs, dys, k, s), js in work:
            B, H, W, cin = xs
            cout = dys[-1]
            if len(js) == 1:
                x, dy, dw, k, s, sc, sh, sl, slc, act = js[0]
                conv_wgrad(x, dy, dw, k, s, in_scale=sc, in_shift=sh, in_slope=sl, in_slope_const=slc, in_act=act)
                continue
            dev = js[0][0].device
            nch = _abi.lib().sst_conv_wgrad_chunks2(B, H, W, cin, cout, k, s, len(js))
            per = nch * k * k * cout * cin
            key = (dev, xs, dys, k, s, len(js))
            ent = WgradGroup._cache.get(key)
            if ent is None:
                ent = {"slab": torch.empty(len(js) * per, device=dev, dtype=torch.float32)}
                WgradGroup._cache[key] = ent
            slab = ent["slab"]
            rows = []
            for i, (x, dy, dw, _, _, sc, sh, sl, slc, act) in enumerate(js):
                bits = struct.unpack("<q", struct.pack("<fi", slc, act))[0]
                rows.append([x.data_ptr(), dy.data_ptr(), slab.data_ptr() + 4 * i * per, dw.data_ptr(), sc.data_ptr() if sc is not None else 0,
                             sh.data_ptr() if sh is not None else 0, sl.data_ptr() if sl is not None else 0, bits])
            # the job table travels to the kernels by value (kernel argument): a host array is all that is needed
            import ctypes
            flatw = [w for r in rows for w in r]
            arr = (ctypes.c_longlong * len(flatw))(*flatw)
            e0 = _prof_begin()
            args = (ctypes.cast(arr, ctypes.c_void_p), len(js), B, H, W, cin, cout, s, k, 0)
            check(_abi.lib().sst_conv_wgrad_grouped(*args, stream_ptr()), "sst_conv_wgrad_grouped")
            flops = 2.0 * B * dys[1] * dys[2] * cout * cin * k * k * len(js)
            name = (_wgrad_name(B, H, W, cin, cout, k, s, len(js)) + "(grouped)+wgrad_reduce_kernel") if (PROFILE is not None or TRACE is not None) else ""
            _prof_end(e0, name, flops)
            _trace(name, flops,
                   lambda args=args: _abi.lib().sst_conv_wgrad_grouped(*args, stream_ptr()), js, slab, arr)
        self.jobs = []


_STAMPS = {}


def debug_stamp(slot):
    """Dev (SST_STAMP=1, tools/stamp_step.py): the device wall clock at this point of the current stream -> debug_stamps()[slot]."""
    if os.environ.get("SST_STAMP", "0") == "0":
        return
    dev = torch.cuda.current_device()
    if dev not in _STAMPS:
        _STAMPS[dev] = torch.zeros(64, dtype=torch.int64, device="cuda")
    check(_abi.lib().sst_debug_stamp(ptr(_STAMPS[dev]), int(slot), stream_ptr()), "sst_debug_stamp")


def debug_stamps():
    return _STAMPS.get(torch.cuda.current_device())

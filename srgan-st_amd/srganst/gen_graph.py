"""Kernel graph of the SRResNet generator: forward and hand-written backward over srganst.ops.

Forward  = reference model.py:142-152 (+155-184), backward = what autograd derives from it.
Activations are NHWC; the NCHW surface is met by the first (transpose of the 3-channel LR input)
and last (conv3 stores NCHW + clamp) kernels.  Fusions relative to the reference's op list:
  * BatchNorm-apply + PReLU of a producer are applied while the consumer conv stages its input;
  * BatchNorm batch statistics are emitted by the conv epilogue (per-tile partials);
  * PixelShuffle is the up-conv's store pattern; clamp_ is conv3's store;
  * PReLU outputs are never materialised (pre-activation tensors are kept, the slope is re-applied on load).
"""
from __future__ import annotations

import torch

from . import ops
from .ops import ACT_SLOPE, OUT_NCHW_CLAMP, OUT_SHUFFLE


class _P:
    """name -> tensor view of the flat parameter / gradient lists of one Generator."""

    def __init__(self, module, tensors):
        self.t = dict(zip(module._names, tensors))

    def __getitem__(self, k):
        return self.t[k]


def _bn_buffers(bn, training):
    return (bn.running_mean, bn.running_var) if training else (None, None)


def _fast9(module, p):
    """The 9x9 convs take the (kx, 3ch)-folded kernels when their small side has exactly 3 channels (the reference's
    RGB configuration) and C is a multiple of 4 that the wgrad kernel supports."""
    C = p["conv1.0.weight"].shape[0]
    return p["conv1.0.weight"].shape[1] == 3 and p["conv3.weight"].shape[0] == 3 and ops.wgrad_c3_supported(C)


def _conv_names(module, fast9):
    names = [] if fast9 else ["conv1.0.weight"]
    for i in range(len(module.trunk)):
        names += [f"trunk.{i}.rcb.0.weight", f"trunk.{i}.rcb.3.weight"]
    names.append("conv2.0.weight")
    names += [f"upsampling.{j}.upsample_block.0.weight" for j in range(len(module.upsampling))]
    if not fast9:
        names.append("conv3.weight")
    return names


def _packs(module, p, with_dgrad, extras=()):
    """All packed weights of the generator from ONE multi-tensor launch: (forward packs by name, data-gradient packs by name or
    None, packs of the (kx, 3ch)-folded 9x9 kernels {"conv1", "conv3", "conv3.dgrad"}).  The data-gradient packs are made by
    the training forward and handed to backward() (the weights do not change in between).  extras: ops.PackPlan's riders."""
    cache = module.__dict__.setdefault("_hip_cache", {})
    fast9 = _fast9(module, p)
    names = _conv_names(module, fast9)
    ws = [p[n] for n in names]
    modes = [ops.PACK_FWD] * len(ws)
    if with_dgrad:
        ws += [p[n] for n in names]
        modes += [ops.PACK_DGRAD] * len(names)
    nine = []
    if fast9:
        nine = [("conv1", "conv1.0.weight", ops.PACK_C3_FWD), ("conv3", "conv3.weight", ops.PACK_TO3)]
        if with_dgrad:
            nine.append(("conv3.dgrad", "conv3.weight", ops.PACK_C3_DGRAD))
        ws += [p[n] for _, n, _ in nine]
        modes += [m for _, _, m in nine]
    out = ops.packed_weights(cache, ("pack", bool(with_dgrad), tuple(e[0] for e in extras)), ws, modes, extras)
    k = len(names)
    wp = dict(zip(names, out[:k]))
    wd = dict(zip(names, out[k:2 * k])) if with_dgrad else None
    w9 = {key: t for (key, _, _), t in zip(nine, out[(2 * k if with_dgrad else k):])}
    return wp, wd, w9


def _bn_accumulators(module, nblocks, C, device, zero, whole=False):
    """The fp64 statistics accumulators of the accumulator mode, forward [2*nblocks+1][NREP][C][2] and backward
    [2*nblocks][NREP][C][4], as views of ONE buffer so that one fill at the start of the forward clears both (whole: that buffer)."""
    nf, nbk = (2 * nblocks + 1) * ops.ACC_NREP * C * 2, 2 * nblocks * ops.ACC_NREP * C * 4
    buf = module.__dict__.get("_bn_acc_buf")
    if buf is None or buf.numel() != nf + nbk or buf.device != device:
        buf = module.__dict__["_bn_acc_buf"] = torch.zeros(nf + nbk, device=device, dtype=torch.float64)
    if whole:
        return buf
    if zero == "all":
        buf.zero_()
    elif zero == "bwd":
        buf[nf:].zero_()
    return buf[:nf].view(2 * nblocks + 1, ops.ACC_NREP, C, 2), buf[nf:].view(2 * nblocks, ops.ACC_NREP, C, 4)


def forward(module, x, params, need_grad):
    p = _P(module, params)
    training = module.training
    sv = {}                                       # saved for backward
    C = p["conv1.0.weight"].shape[0]
    # the batch counters' add and the clearing of the statistics accumulators (two one-line launches at the head of the serial chain)
    # ride in the pack launch
    B_, H_, W_ = x.shape[0], x.shape[2], x.shape[3]
    use_acc = bool(training and len(module.trunk) and ops.conv_acc_supported(B_, H_, W_, C, C))
    extras = []
    counters = ops.flatten_bn_counters(module) if training else None
    if counters is not None:
        extras.append(("add", counters, 1))
    if use_acc:
        extras.append(("zero", _bn_accumulators(module, len(module.trunk), C, x.device, zero=None, whole=True)))
    wp, sv["wd"], w9 = _packs(module, p, need_grad, tuple(extras))
    sv["w9"] = w9
    x3 = ops.transpose(x.contiguous(), to_nchw=False)                                  # [B,h,w,3]
    fast9 = _fast9(module, p)
    if fast9:
        z1 = ops.conv9_c3_fwd(x3, p["conv1.0.weight"], 0, bias=p["conv1.0.bias"], wp=w9["conv1"])
    else:
        z1, _, _, _ = ops.conv_fwd(x3, wp["conv1.0.weight"], C, 9, 1, bias=p["conv1.0.bias"])
    a1 = p["conv1.1.weight"]
    sv["x3"], sv["z1"] = x3, z1

    def bn_affine(bn, pre, stats, cnt):
        if training:
            mean, rstd, scale, shift = ops.bn_finalize(stats, cnt, p[pre + ".weight"], p[pre + ".bias"],
                                                       bn.running_mean, bn.running_var)
            return mean, rstd, scale, shift
        scale, shift = ops.bn_eval_affine(p[pre + ".weight"], p[pre + ".bias"], bn.running_mean, bn.running_var)
        return None, None, scale, shift

    h = z1
    blocks = []
    n_px = float(B_ * H_ * W_)
    if use_acc:
        # ---- accumulator mode: no BatchNorm finalize launches - each conv adds its output statistics into fp64 accumulators and
        # the NEXT conv derives the affine of its input from them in its prologue (csrc/conv_epilogue.h: BandAcc)
        nbn = 2 * len(module.trunk) + 1
        acc, _ = _bn_accumulators(module, len(module.trunk), C, z1.device, zero=None)       # cleared by the pack launch above
        sv["acc_token"] = module.__dict__["_bn_acc_token"] = object()      # backward: its accumulators are still clear

        def stat_tensors():
            return tuple(ops._f32(C, like=z1) for _ in range(4))

        pend = None       # (x, y2, acc2, bn2 module, prefix, (m2, r2, s2, t2)): residual sum formed by the next conv
        for i, blk in enumerate(module.trunk):
            pre = f"trunk.{i}.rcb"
            first = i == 0
            a_k1, a_k2 = acc[2 * i], acc[2 * i + 1]
            if pend is None:
                y1, _ = ops.conv_fwd_acc(h, wp[pre + ".0.weight"], C, 3, in_slope=a1 if first else None,
                                         in_act=ACT_SLOPE if first else 0, st_acc=a_k1)
            else:
                px, py2, pacc, pbn, ppre, pst = pend
                y1, h = ops.conv_fwd_acc(px, wp[pre + ".0.weight"], C, 3, in2=py2, in_acc=pacc, in_bn=(p[ppre + ".weight"], p[ppre + ".bias"]),
                                         n=n_px, out_stats=pst, run_stats=(pbn.running_mean, pbn.running_var), st_acc=a_k1)
            st1 = stat_tensors()
            y2, _ = ops.conv_fwd_acc(y1, wp[pre + ".3.weight"], C, 3, in_slope=p[pre + ".2.weight"], in_act=ACT_SLOPE, in_acc=a_k1,
                                     in_bn=(p[pre + ".1.weight"], p[pre + ".1.bias"]), n=n_px, out_stats=st1,
                                     run_stats=(blk.rcb[1].running_mean, blk.rcb[1].running_var), st_acc=a_k2)
            if first:     # the skip term is PReLU(z1): stand-alone finalize + residual kernel for this one block
                st2 = ops.bn_finalize_acc(a_k2, n_px, p[pre + ".4.weight"], p[pre + ".4.bias"], blk.rcb[4].running_mean,
                                          blk.rcb[4].running_var)
                blocks.append((h, y1, *st1, y2, *st2))
                h, pend = ops.bn_residual(y2, st2[2], st2[3], h, a1), None
            else:
                st2 = stat_tensors()
                blocks.append((h, y1, *st1, y2, *st2))
                pend = (h, y2, a_k2, blk.rcb[4], pre + ".4", st2)
        a_k3 = acc[nbn - 1]
        if pend is None:
            y3, _ = ops.conv_fwd_acc(h, wp["conv2.0.weight"], C, 3, st_acc=a_k3)
        else:
            px, py2, pacc, pbn, ppre, pst = pend
            y3, h = ops.conv_fwd_acc(px, wp["conv2.0.weight"], C, 3, in2=py2, in_acc=pacc, in_bn=(p[ppre + ".weight"], p[ppre + ".bias"]),
                                     n=n_px, out_stats=pst, run_stats=(pbn.running_mean, pbn.running_var), st_acc=a_k3)
        m3, r3, s3, t3 = ops.bn_finalize_acc(a_k3, n_px, p["conv2.1.weight"], p["conv2.1.bias"], module.conv2[1].running_mean,
                                             module.conv2[1].running_var)
    else:
        h = z1
        blocks = []
        pend = None       # (x, y2, scale2, shift2): the previous block's output h = x + BN2(y2), not materialised yet - the next
                          # conv forms it while staging its input and hands it back (one bn_residual launch less per block)
        for i, blk in enumerate(module.trunk):
            pre = f"trunk.{i}.rcb"
            first = i == 0
            if pend is None:
                y1, _, st, cnt = ops.conv_fwd(h, wp[pre + ".0.weight"], C, 3, 1,
                                              in_slope=a1 if first else None, in_act=ACT_SLOPE if first else 0,
                                              want_stats=training)
            else:
                y1, h, st, cnt = ops.conv_fwd_resin(*pend, wp[pre + ".0.weight"], C, 3, want_stats=training)
            m1, r1, s1, t1 = bn_affine(blk.rcb[1], pre + ".1", st, cnt)
            y2, _, st, cnt = ops.conv_fwd(y1, wp[pre + ".3.weight"], C, 3, 1, in_scale=s1, in_shift=t1,
                                          in_slope=p[pre + ".2.weight"], in_act=ACT_SLOPE, want_stats=training)
            m2, r2, s2, t2 = bn_affine(blk.rcb[4], pre + ".4", st, cnt)
            blocks.append((h, y1, m1, r1, s1, t1, y2, m2, r2, s2, t2))
            if first:     # the skip term is PReLU(z1) here: keep the stand-alone residual kernel
                h, pend = ops.bn_residual(y2, s2, t2, h, a1), None
            else:
                pend = (h, y2, s2, t2)
        if pend is None:
            y3, _, st, cnt = ops.conv_fwd(h, wp["conv2.0.weight"], C, 3, 1, want_stats=training)
        else:
            y3, h, st, cnt = ops.conv_fwd_resin(*pend, wp["conv2.0.weight"], C, 3, want_stats=training)
        m3, r3, s3, t3 = bn_affine(module.conv2[1], "conv2.1", st, cnt)
    u = ops.bn_residual(y3, s3, t3, z1, a1)
    sv["blocks"], sv["h_last"], sv["conv2"] = blocks, h, (y3, m3, r3, s3, t3)
    ups = []
    slope = None
    for j, _ in enumerate(module.upsampling):
        pre = f"upsampling.{j}.upsample_block"
        us, _, _, _ = ops.conv_fwd(u, wp[pre + ".0.weight"], 4 * C, 3, 1, bias=p[pre + ".0.bias"],
                                   in_slope=slope, in_act=ACT_SLOPE if slope is not None else 0, out_mode=OUT_SHUFFLE)
        ups.append((u, slope, us))
        u, slope = us, p[pre + ".2.weight"]
    sv["ups"] = ups
    cout = p["conv3.weight"].shape[0]
    if fast9:
        sr, sr_pre = ops.conv9_to3_fwd(u, p["conv3.weight"], bias=p["conv3.bias"], in_slope=slope,
                                       in_act=ACT_SLOPE if slope is not None else 0, want_pre=need_grad, wp=w9["conv3"])
    else:
        sr, sr_pre, _, _ = ops.conv_fwd(u, wp["conv3.weight"], cout, 9, 1, bias=p["conv3.bias"], in_slope=slope,
                                        in_act=ACT_SLOPE if slope is not None else 0, out_mode=OUT_NCHW_CLAMP,
                                        want_pre=need_grad)
    sv["last"] = (u, slope, sr_pre)
    return sr, sv


def backward(module, params, sv, dsr, need_dx=False):
    """-> list of gradients in module._names order."""
    p = _P(module, params)
    grads = ops.flat_grads(module, module._names, params)      # views of ONE flat buffer (single RCCL message)
    C = p["conv1.0.weight"].shape[0]
    a1 = p["conv1.1.weight"]
    wd, w9 = sv["wd"], sv["w9"]                  # packed by the forward's multi-tensor launch
    wg = ops.WgradGroup()            # the trunk-shaped weight gradients go out as ONE launch at the end
    # Weight gradients are leaves of the backward chain.  An owner that runs another branch beside this backward (engine.TrainEngine's
    # merged iteration: the discriminator step on a side stream) may take them over: module._defer_wgrad = (list, mask) - the launches
    # selected by the mask (1 conv3, 2 up-sampling convs, 4 the grouped trunk launch, 8 conv1) are appended as (event recorded on this
    # stream once their operands exist, launch closure, tensors) instead of issued; same kernels, same arguments.
    dlist, dmask = module.__dict__.get("_defer_wgrad") or (None, 0)

    def leaf(bit, launch, *tensors):
        if dlist is not None and (dmask & bit):
            ev = torch.cuda.Event()
            ev.record()
            dlist.append((ev, launch, tensors))
        else:
            with ops.SideStream(*tensors):
                launch()

    def rows(t):
        return t.numel() // t.shape[-1]

    # ---- conv3 + clamp
    u, slope, sr_pre = sv["last"]
    g3 = ops.clamp_bwd(dsr.contiguous(), sr_pre, dbias=grads["conv3.bias"])
    fast9 = _fast9(module, p)
    def w_conv3(u=u, g3=g3, slope=slope):
        if fast9:
            ops.wgrad_c3(u, g3, grads["conv3.weight"], 0, in_slope=slope, in_act=ACT_SLOPE if slope is not None else 0)
        else:
            ops.conv_wgrad(u, g3, grads["conv3.weight"], 9, 1, in_slope=slope, in_act=ACT_SLOPE if slope is not None else 0)
    leaf(1, w_conv3, u, g3, grads["conv3.weight"])
    if fast9:
        g = ops.conv9_c3_fwd(g3, p["conv3.weight"], 1, wp=w9["conv3.dgrad"])              # d PReLU(u)
    else:
        g = ops.conv_fwd(g3, wd["conv3.weight"], C, 9, 1)[0]
    # ---- up-sampling blocks, last to first
    for j in reversed(range(len(sv["ups"]))):
        pre = f"upsampling.{j}.upsample_block"
        u_in, slope_in, us = sv["ups"][j]
        sl = p[pre + ".2.weight"]
        # PReLU backward + inverse PixelShuffle + the partial sums of (conv bias, slope) gradients in ONE pass, one finalize
        du = ops.act_bwd(g, us, slope=sl, dbias=grads[pre + ".0.bias"], dslope=grads[pre + ".2.weight"],
                         unshuffle=True)                                             # [B,h,w,4C] pre-shuffle grad
        def w_up(u_in=u_in, du=du, pre=pre, slope_in=slope_in):
            ops.conv_wgrad(u_in, du, grads[pre + ".0.weight"], 3, 1, in_slope=slope_in,
                           in_act=ACT_SLOPE if slope_in is not None else 0)
        leaf(2, w_up, u_in, du, grads[pre + ".0.weight"])
        g = ops.conv_fwd(du, wd[pre + ".0.weight"], C, 3, 1)[0]    # d (input of the up-conv)
    # ---- u = BN(conv2(h_last)) + PReLU(z1)
    y3, m3, r3, s3, t3 = sv["conv2"]
    n = rows(y3)
    dy3 = ops.bwd_reduce_apply(g, y3, n, scale=s3, shift=t3, mean=m3, rstd=r3, gamma=p["conv2.1.weight"],
                               dgamma=grads["conv2.1.weight"], dbeta=grads["conv2.1.bias"])
    wg.add(sv["h_last"], dy3, grads["conv2.0.weight"], 3, 1,
           in_slope=a1 if not sv["blocks"] else None, in_act=ACT_SLOPE if not sv["blocks"] else 0)
    dskip = g
    nb = len(sv["blocks"])
    Bq, Hq, Wq, _ = y3.shape
    if nb and ops.conv_acc_supported(Bq, Hq, Wq, C, C):
        # ---- accumulator mode: every data-gradient conv adds the BatchNorm-backward sums of its result into fp64 accumulators,
        # the next stage derives its coefficients from them in its prologue (no finalize launches between the stages)
        _, bacc = _bn_accumulators(module, nb, C, y3.device, zero="bwd" if sv.get("acc_token") is None or
                                   module.__dict__.get("_bn_acc_token") is not sv.get("acc_token") else None)
        module.__dict__["_bn_acc_token"] = None          # the backward accumulators are dirty from here on
        dh, _ = ops.conv_dgrad_fused_acc(dy3, wd["conv2.0.weight"], C, 3, epi_y=sv["blocks"][-1][6], bw_st_acc=bacc[2 * nb - 1])
        for i in reversed(range(nb)):
            pre = f"trunk.{i}.rcb"
            first = i == 0
            h, y1, m1, r1, s1, t1, y2, m2, r2, s2, t2 = sv["blocks"][i]
            sl = p[pre + ".2.weight"]
            # stage 2 (BN2, no activation): [coefficients + apply + dgrad conv_b + sums for stage 1] in one launch
            dp1, dy2 = ops.conv_dgrad_fused_acc(dh, wd[pre + ".3.weight"], C, 3, y2=y2, epi_y=y1, epi_scale=s1, epi_shift=t1, epi_slope=sl,
                                                epi_act=1, bw_in_acc=bacc[2 * i + 1], bn=(m2, r2, p[pre + ".4.weight"]), n=n,
                                                dgamma=grads[pre + ".4.weight"], dbeta=grads[pre + ".4.bias"], bw_st_acc=bacc[2 * i])
            wg.add(y1, dy2, grads[pre + ".3.weight"], 3, 1, in_scale=s1, in_shift=t1, in_slope=sl, in_act=ACT_SLOPE)
            # stage 1 (BN1 + PReLU)
            prev_y2 = None if first else sv["blocks"][i - 1][6]
            dh, dy1 = ops.conv_dgrad_fused_acc(dp1, wd[pre + ".0.weight"], C, 3, y2=y1, in_scale=s1, in_shift=t1, in_slope=sl,
                                               in_act=ACT_SLOPE, residual=dh, epi_y=prev_y2, bw_in_acc=bacc[2 * i],
                                               bn=(m1, r1, p[pre + ".1.weight"]), n=n, dgamma=grads[pre + ".1.weight"],
                                               dbeta=grads[pre + ".1.bias"], dslope=grads[pre + ".2.weight"],
                                               bw_st_acc=None if first else bacc[2 * i - 1])
            wg.add(h, dy1, grads[pre + ".0.weight"], 3, 1, in_slope=a1 if first else None, in_act=ACT_SLOPE if first else 0)
    else:
        # every stride-1 data-gradient conv below also emits the BatchNorm-backward partial sums of its result against the
        # conv output the NEXT backward stage differentiates through (saves one reduction pass per stage)
        if nb:
            dh, part = ops.conv_dgrad_bwdstats(dy3, wd["conv2.0.weight"], C, 3, sv["blocks"][-1][6])        # vs y2 of the last block
        else:
            dh, part = ops.conv_fwd(dy3, wd["conv2.0.weight"], C, 3, 1)[0], None
        # ---- residual blocks, last to first
        for i in reversed(range(nb)):
            pre = f"trunk.{i}.rcb"
            first = i == 0
            h, y1, m1, r1, s1, t1, y2, m2, r2, s2, t2 = sv["blocks"][i]
            sl = p[pre + ".2.weight"]
            # stage 2 (BN2, no activation): finalize -> [apply + dgrad conv_b + partials for stage 1] in one launch
            cA, cB, cC = ops.bwd_finalize(part, n, m2, r2, p[pre + ".4.weight"], grads[pre + ".4.weight"], grads[pre + ".4.bias"])
            dp1, dy2, part = ops.conv_dgrad_fused(dh, y2, wd[pre + ".3.weight"], C, 3, cA, cB, cC, epi_y=y1, epi_scale=s1,
                                                  epi_shift=t1, epi_slope=sl, epi_act=1)
            wg.add(y1, dy2, grads[pre + ".3.weight"], 3, 1, in_scale=s1, in_shift=t1, in_slope=sl, in_act=ACT_SLOPE)
            # stage 1 (BN1 + PReLU)
            cA, cB, cC = ops.bwd_finalize(part, n, m1, r1, p[pre + ".1.weight"], grads[pre + ".1.weight"], grads[pre + ".1.bias"],
                                          dslope=grads[pre + ".2.weight"])
            prev_y2 = None if first else sv["blocks"][i - 1][6]
            dh, dy1, part = ops.conv_dgrad_fused(dp1, y1, wd[pre + ".0.weight"], C, 3, cA, cB, cC, in_scale=s1, in_shift=t1,
                                                 in_slope=sl, in_act=ACT_SLOPE, residual=dh, epi_y=prev_y2)
            wg.add(h, dy1, grads[pre + ".0.weight"], 3, 1, in_slope=a1 if first else None, in_act=ACT_SLOPE if first else 0)
    # ---- c1 = PReLU(z1): gradient = trunk path (dh) + global skip (dskip)
    z1 = sv["z1"]
    dz1 = ops.act_bwd(dh, z1, g2=dskip, slope=a1, dbias=grads["conv1.0.bias"], dslope=grads["conv1.1.weight"])
    def w_conv1(dz1=dz1, x3=sv["x3"]):
        if fast9:
            ops.wgrad_c3(dz1, x3, grads["conv1.0.weight"], 1)
        else:
            ops.conv_wgrad(x3, dz1, grads["conv1.0.weight"], 9, 1)
    leaf(8, w_conv1, dz1, sv["x3"], grads["conv1.0.weight"])
    if dlist is not None and (dmask & 4) and wg.jobs:
        ev = torch.cuda.Event()
        ev.record()
        dlist.append((ev, wg.run, tuple(t for j in wg.jobs for t in j[:3])))
    else:
        wg.run()
    ops.join_side()
    dx = None
    if need_dx:
        wd1 = ops.pack_conv(p["conv1.0.weight"], 1) if fast9 else wd["conv1.0.weight"]
        dx3 = ops.conv_fwd(dz1, wd1, sv["x3"].shape[-1], 9, 1)[0]
        dx = ops.transpose(dx3, to_nchw=True)
    return [grads[n] for n in module._names], dx


class GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, grad_mode, *params):
        need_grad = grad_mode and any(ctx.needs_input_grad)
        if need_grad and not module.training:
            raise NotImplementedError("Generator backward in eval() mode is not on the reference's path "
                                      "(train.py:109 / warmup.py:71 call .train() before every epoch)")
        sr, sv = forward(module, x, [t.detach() for t in params], need_grad)
        if need_grad:
            ctx.module, ctx.sv, ctx.params = module, sv, [t.detach() for t in params]
            ctx.need_dx = ctx.needs_input_grad[0]
        return sr

    @staticmethod
    def backward(ctx, dsr):
        grads, dx = backward(ctx.module, ctx.params, ctx.sv, dsr, ctx.need_dx)
        ctx.sv = None
        return (dx, None, None, *grads)

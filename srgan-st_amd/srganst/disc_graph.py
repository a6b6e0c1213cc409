"""Kernel graph of the VGG-style discriminator: forward and hand-written backward over srganst.ops.

Forward = reference model.py:67-71 over the layer list model.py:30-65; backward = what autograd
derives.  NHWC activations; conv outputs are kept pre-BatchNorm / pre-activation and the
BN-apply + LeakyReLU(0.2) is fused into the consumer's load.  The flatten (model.py:69) keeps the
reference's (C,H,W) order so classifier.0.weight is used in its reference layout.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import ACT_SLOPE

LRELU = 0.2
# (conv index, bn index or None, stride) in module.features                           model.py:30-59
PLAN = [(0, None, 1), (2, 3, 2), (5, 6, 1), (8, 9, 2), (11, 12, 1), (14, 15, 2), (17, 18, 1), (20, 21, 2)]


def _packs(module, p, with_dgrad, counter_add=0):
    """(forward packs by name, stride-1 data-gradient packs by name, stride-2 data-gradient packs by name) - the last two None
    without with_dgrad.  One multi-tensor launch + one launch per stride-2 layer.

    Re-use: the discriminator runs up to three forward and three backward passes between two updates of its weights
    (train.py:125-161: D(sr) in the generator step, D(gt) and D(sr) in its own step).  An owner that controls those updates
    (engine.TrainEngine) sets module._packs_managed and clears module._packs_fresh after every optimizer step; while the flag
    is set (and no parameter was rebound or modified through torch, see _version) the packed buffers are handed out again
    without a launch.  Without an owner every call packs, as before.
    counter_add > 0 (the owner's call at the head of an iteration, which always packs): the BatchNorm batch counters' add for the
    iteration's passes rides in the pack launch."""
    cache = module.__dict__.setdefault("_hip_cache", {})
    names = [f"features.{ci}.weight" for ci, _, s in PLAN]
    n1 = [f"features.{ci}.weight" for ci, _, s in PLAN if s == 1]
    n2 = [f"features.{ci}.weight" for ci, _, s in PLAN if s == 2]
    sig = tuple((p[n].data_ptr(), p[n]._version) for n in names)
    st = cache.get("pack_state")
    if (module.__dict__.get("_packs_managed") and module.__dict__.get("_packs_fresh") and st is not None and st["sig"] == sig
            and (st["wd"] is not None or not with_dgrad)):
        assert not counter_add, "the packs are fresh: nothing to ride along with"
        return st["wp"], st["wd"], st["ws2"]
    ws = [p[n] for n in names] + ([p[n] for n in n1] if with_dgrad else [])
    modes = [ops.PACK_FWD] * len(names) + ([ops.PACK_DGRAD] * len(n1) if with_dgrad else [])
    extras = (("add", ops.flatten_bn_counters(module), int(counter_add)),) if counter_add else ()
    out = ops.packed_weights(cache, ("pack", bool(with_dgrad), int(counter_add)), ws, modes, extras)
    wp = dict(zip(names, out[:len(names)]))
    wd = ws2 = None
    if with_dgrad:
        wd = dict(zip(n1, out[len(names):]))
        ws2 = {n: ops.pack_conv_s2_dgrad(p[n], out=cache.get(("s2", n))) for n in n2}
        for n in n2:
            cache[("s2", n)] = ws2[n]
    cache["pack_state"] = {"sig": sig, "wp": wp, "wd": wd, "ws2": ws2}
    module.__dict__["_packs_fresh"] = True
    return wp, wd, ws2


def groups_supported(module, p, gB, groups, H, W):
    """Can `groups` passes of gB images each (H x W) run as ONE batch with per-pass BatchNorm statistics?  Every BatchNorm layer's
    producer and consumer must be on the kernels that take coefficient groups (the pipelined conv, the all-taps weight gradient)."""
    B = gB * groups
    h, w = H, W
    for li, (ci, bi, stride) in enumerate(PLAN):
        wt = p[f"features.{ci}.weight"]
        cout, cin = wt.shape[0], wt.shape[1]
        if li > 0:
            if not ops.conv_pipe_groups_ok(B, h, w, cin, cout, 3, stride, gB):
                return False
            if stride == 1 and not ops.conv_pipe_groups_ok(B, h, w, cout, cin, 3, 1, gB):      # its data-gradient (epilogue partials)
                return False
            if not ops._abi.lib().sst_conv_wgrad_groups_ok(B, h, w, cin, cout, 3, stride, gB):
                return False
        h, w = ops.conv_out_hw(h, w, 3, stride)
    return B <= 64            # classifier kernels: at most 64 batch rows


class PassArena:
    """Activations of several passes side by side: every tensor the backward reads is allocated once for `slots` passes of B images
    and each pass's forward (slot = its place) writes its slice.  Passes that ran separately - the generator step's D(sr) and, later,
    the discriminator step's D(gt) (engine.TrainEngine with KERNEL.REUSE_D_SR) - can then be differentiated as ONE batch
    (batched_saved): per-pass BatchNorm rows, one launch per layer over slots * B images."""

    def __init__(self, slots):
        self.slots, self.t, self.filled, self.meta = slots, {}, set(), None

    def get(self, key, shape, like, slot, B):
        t = self.t.get(key)
        if t is None:
            t = self.t[key] = torch.empty(self.slots * shape[0], *shape[1:], device=like.device, dtype=torch.float32)
        return t[slot * B:(slot + 1) * B] if B else t[slot]


def batched_saved(arena, sv_any):
    """The saved-tensor record of `arena.slots` passes as one batch (what forward() on a list of inputs would have saved), once every slot
    has been filled by a forward(..., arena=(arena, slot)) of the same weights."""
    assert len(arena.filled) == arena.slots, "not every pass of the arena has run"
    B = arena.meta["B"]
    sv = {"layers": [], "groups": arena.slots, "gB": B, "wd": sv_any["wd"], "ws2": sv_any["ws2"],
          "flat": arena.t["flat"], "h1": arena.t["h1"]}
    h, scale, shift, act = arena.t["x3"], None, None, 0
    for li, r in enumerate(sv_any["layers"]):
        rec = {"x": h, "x_scale": scale, "x_shift": shift, "x_act": act, "y": arena.t[("y", li)], "ci": r["ci"], "bi": r["bi"],
               "stride": r["stride"]}
        if r["bi"] is not None:
            rec["mean"], rec["rstd"] = arena.t[("mean", li)], arena.t[("rstd", li)]
            scale, shift = arena.t[("scale", li)], arena.t[("shift", li)]
        else:
            scale = shift = None
        rec["scale"], rec["shift"] = scale, shift
        sv["layers"].append(rec)
        h, act = rec["y"], ACT_SLOPE
    return sv


def forward(module, x, p, training, need_grad=False, bump_counters=True, bn_hook=None, update_running=True, arena=None):
    """x: one NCHW batch, or a LIST of `groups` equally-shaped NCHW batches = that many passes of the discriminator run as ONE
    batch (each pass keeps its own train-mode BatchNorm statistics: per-pass scale / shift rows, running statistics updated pass by
    pass in list order, batch counters + groups) - the discriminator step's D(gt) and D(sr.detach()) (train.py:155-158) as one tall
    image; returns logits [groups * B, 1] and saved tensors with sv["groups"], sv["gB"].
    bump_counters=False: the caller has already added this pass to num_batches_tracked (two passes on two streams must
    not race on the counters).  bn_hook(li, when) is called right before ("pre") / after ("post") each train-mode
    bn_finalize - the only kernels of a forward that write shared state (running statistics): a caller that runs two passes
    concurrently orders them there.  update_running=False: the pass leaves the running statistics and the batch counter alone;
    the caller applies them later, in the reference's order, with replay_running_stats (a pass that runs EARLIER than its place
    in the reference's sequence)."""
    if not update_running:
        bump_counters = False
    groups, gB = 1, 0
    ar, slot = arena if arena is not None else (None, 0)      # (PassArena, slot): this pass writes its slice of the shared tensors
    if ar is not None:
        assert training and not isinstance(x, (list, tuple))
        Bx = x.shape[0]
        if ar.meta is None:
            ar.meta = {"B": Bx}
        assert ar.meta["B"] == Bx

    def A(key, shape, like):
        return ar.get(key, shape, like, slot, shape[0]) if ar is not None else None
    if isinstance(x, (list, tuple)):
        groups, gB = len(x), x[0].shape[0]
        if groups == 1:
            x, gB = x[0], 0
        elif not training:
            raise NotImplementedError("batched passes are a train-mode schedule (eval mode has no per-pass statistics to keep apart)")
    sv = {"layers": [], "groups": groups, "gB": gB}
    wp, sv["wd"], sv["ws2"] = _packs(module, p, need_grad)
    if training and bump_counters and not module.__dict__.get("_counters_external"):
        ops.flatten_bn_counters(module).add_(groups)
    if groups > 1:
        _, c3, hh, ww = x[0].shape
        x3 = torch.empty(groups * gB, hh, ww, c3, device=x[0].device, dtype=torch.float32)
        for gi, xi in enumerate(x):
            assert xi.shape == x[0].shape
            ops.transpose(xi.contiguous(), to_nchw=False, out=x3[gi * gB:(gi + 1) * gB])
    else:
        Bq, c3, hh, ww = x.shape
        x3 = ops.transpose(x.contiguous(), to_nchw=False, out=A("x3", (Bq, hh, ww, c3), x))
    h, scale, shift, act = x3, None, None, 0
    for ci, bi, stride in PLAN:
        w = p[f"features.{ci}.weight"]
        cout = w.shape[0]
        bias = p.get(f"features.{ci}.bias")
        ho, wo = ops.conv_out_hw(h.shape[1], h.shape[2], 3, stride)
        y, _, st, cnt = ops.conv_fwd(h, wp[f"features.{ci}.weight"], cout, 3, stride, bias=bias, in_scale=scale, in_shift=shift,
                                     in_slope_const=LRELU, in_act=act, want_stats=(bi is not None and training), grp=gB,
                                     out=A(("y", len(sv["layers"])), (h.shape[0], ho, wo, cout), h))
        rec = {"x": h, "x_scale": scale, "x_shift": shift, "x_act": act, "y": y, "ci": ci, "bi": bi, "stride": stride}
        if bi is not None:
            bn = module.features[bi]
            g, b = p[f"features.{bi}.weight"], p[f"features.{bi}.bias"]
            if training:
                if bn_hook is not None:
                    bn_hook(len(sv["layers"]), "pre")
                rows = None
                if ar is not None:      # this pass's row of the [passes, C] coefficient tables
                    li_ = len(sv["layers"])
                    rows = tuple(ar.get((k, li_), (1, cout), h, slot, 0) for k in ("mean", "rstd", "scale", "shift"))
                mean, rstd, scale, shift = ops.bn_finalize(st, cnt, g, b, bn.running_mean if update_running else None,
                                                           bn.running_var if update_running else None, groups=groups, out=rows)
                if bn_hook is not None:
                    bn_hook(len(sv["layers"]), "post")
                rec["mean"], rec["rstd"] = mean, rstd
                rec["st"], rec["cnt"] = st, cnt             # replay_running_stats
            else:
                scale, shift = ops.bn_eval_affine(g, b, bn.running_mean, bn.running_var)
        else:
            scale = shift = None
        rec["scale"], rec["shift"] = scale, shift
        sv["layers"].append(rec)
        h, act = y, ACT_SLOPE
    nfeat = h.shape[1] * h.shape[2] * h.shape[3]
    flat = ops.flatten_act(h, scale, shift, LRELU, 1, grp=gB, out=A("flat", (h.shape[0], nfeat), h))      # [B, C*H*W]  (C,H,W) order
    h1 = ops.linear_fwd(flat, p["classifier.0.weight"], p["classifier.0.bias"],
                        out=A("h1", (h.shape[0], p["classifier.0.weight"].shape[0]), h))   # pre-activation
    if ar is not None:
        ar.filled.add(slot)
    out = ops.head_fwd(h1, p["classifier.2.weight"], p["classifier.2.bias"], LRELU)
    sv["flat"], sv["h1"] = flat, h1
    return out, sv


def replay_running_stats(module, p, sv):
    """The side effects of ONE MORE train-mode forward over the input and weights of the pass saved in `sv`, without running it:
    every kernel of the forward is deterministic, so the second pass would reproduce the saved activations bit for bit - all it
    would add is one more step of the BatchNorm running statistics (same batch statistics: bn_finalize again on the saved
    per-tile partials, i.e. the very launch the forward would issue) and of num_batches_tracked.  engine.TrainEngine uses it for
    the discriminator step's D(sr.detach()) (train.py:158), which repeats the generator step's D(sr) (train.py:136) before any
    weight has changed."""
    groups = sv.get("groups", 1)
    if not module.__dict__.get("_counters_external"):
        ops.flatten_bn_counters(module).add_(groups)
    for rec in sv["layers"]:
        if rec["bi"] is None:
            continue
        bn = module.features[rec["bi"]]
        ops.bn_finalize(rec["st"], rec["cnt"], p[f"features.{rec['bi']}.weight"], p[f"features.{rec['bi']}.bias"],
                        bn.running_mean, bn.running_var, groups=groups)


def _grad_views(module, p, need_param_grads):
    """Per-parameter gradient views of ONE flat buffer + whether this pass accumulates into it.
    Two backward passes per D step (D(gt) and D(sr), train.py:155-161): when the step engine opened an accumulation scope
    (module._grad_accum), the second pass ADDS into the first pass's flat buffer with the kernels' accumulate flag and hands
    autograd nothing - p.grad stays a view of ONE flat buffer (flat Adam, single RCCL message) and autograd's own
    out-of-place sum of two 94 MB gradient sets disappears."""
    names = list(p.keys())
    scope = module.__dict__.get("_grad_accum") if need_param_grads else None
    acc = scope is not None and scope.get("flat") is not None
    if acc:
        plist = [p[n] for n in names]
        offs, _ = ops.flat_layout(plist)
        views = {n: scope["flat"][o:o + t.numel()].view(t.shape) for n, t, o in zip(names, plist, offs)}
    else:
        views = ops.flat_grads(module, names, [p[n] for n in names]) if need_param_grads else {}
        if scope is not None:
            scope["flat"] = module.__dict__["_flat_grads"][-1]
    return views, acc


def backward_classifier(module, p, sv, dout, need_param_grads, st=None):
    """Backward of the classifier (model.py:61-65) of one pass: fills classifier.* gradients, returns the state the
    feature-stack half needs (incl. g = d LeakyReLU(BN(y_last)) in NHWC).  The data-parallel engine runs this half of both
    passes first, so that the classifier bucket (75.5 MB) can be all-reduced while the feature stack's backward runs."""
    views, acc = _grad_views(module, p, need_param_grads)
    grads = {}

    def G(name):
        t = views[name]
        grads[name] = t
        return t

    h1, flat = sv["h1"], sv["flat"]
    wg = need_param_grads
    dh1 = ops.head_bwd(h1, p["classifier.2.weight"], dout.contiguous(), LRELU,
                       dw=G("classifier.2.weight") if wg else None, db=G("classifier.2.bias") if wg else None, accumulate=acc)
    if wg:
        dw0, db0 = G("classifier.0.weight"), G("classifier.0.bias")
        with ops.SideStream(dh1, flat, dw0, db0):
            ops.linear_wgrad(dh1, flat, dw0, db0, accumulate=acc)
    last = sv["layers"][-1]
    B, H, W, C = last["y"].shape
    g = ops.linear_dgrad(dh1, p["classifier.0.weight"], nhwc=(C, H * W)).view(B, H, W, C)   # d LReLU(BN(y_last)) in NHWC
    return {"views": views, "acc": acc, "grads": grads, "g": g}


def backward_features(module, p, sv, st, need_param_grads, need_dx, defer_wgrad=None, defer_below=None):
    """Backward of the feature stack (model.py:30-59) of one pass, continuing from backward_classifier's state.
    defer_wgrad (a list): the conv weight gradients are not launched here - they are leaves of the backward chain (nothing in this
    pass reads them) - but appended as (event recorded on the current stream once dy exists, launch closure); the caller runs them
    wherever the chip has room (engine.TrainEngine._iter_gd: on the generator's stream once its backward is done).  defer_below: only
    the layers with index < defer_below (the ones the chain reaches last) are deferred."""
    views, acc, grads, g = st["views"], st["acc"], st["grads"], st["g"]

    def G(name):
        t = views[name]
        grads[name] = t
        return t

    wg = need_param_grads
    dx = None
    groups, gB = sv.get("groups", 1), sv.get("gB", 0)      # passes batched as one tall image: per-pass coefficient rows
    wd, ws2 = sv["wd"], sv["ws2"]               # packed (or re-used) by the forward
    part = None                      # BN/activation backward partials of g, when the producing dgrad conv emitted them
    reduces = [] if ops.WGRAD_REDUCE_MULTI else None      # slab reduces left for one launch at the end of the pass
    for li in reversed(range(len(sv["layers"]))):
        r = sv["layers"][li]
        y = r["y"]
        n = y.numel() // y.shape[-1] // groups      # elements per channel of ONE pass (each has its own batch statistics)
        ci, bi = r["ci"], r["bi"]
        w = p[f"features.{ci}.weight"]
        if bi is not None:
            gam = p[f"features.{bi}.weight"]
            dg = G(f"features.{bi}.weight") if wg else torch.empty_like(gam)
            db = G(f"features.{bi}.bias") if wg else torch.empty_like(gam)
            kw = dict(scale=r["scale"], shift=r["shift"], slope_const=LRELU, act=1, mean=r["mean"], rstd=r["rstd"], gamma=gam,
                      dgamma=dg, dbeta=db, accumulate=acc and wg, groups=groups)
            dy = ops.bwd_finalize_apply(part, g, y, n, **kw) if part is not None else ops.bwd_reduce_apply(g, y, n, **kw)
        elif wg:
            kw = dict(slope_const=LRELU, act=1, dbeta=G(f"features.{ci}.bias"), accumulate=acc, groups=groups)
            dy = ops.bwd_finalize_apply(part, g, y, n, **kw) if part is not None else ops.bwd_reduce_apply(g, y, n, **kw)
        else:
            dy = ops.bwd_apply(g, y, slope_const=LRELU, act=1, groups=groups)
        part = None
        if wg:
            dwc = G(f"features.{ci}.weight")

            def launch(r=r, dy=dy, dwc=dwc, red=None):
                ops.conv_wgrad(r["x"], dy, dwc, 3, r["stride"], in_scale=r["x_scale"], in_shift=r["x_shift"],
                               in_slope_const=LRELU, in_act=r["x_act"], accumulate=acc, grp=gB, defer_reduce=red)
            if defer_wgrad is not None and (defer_below is None or li < defer_below):
                ev = torch.cuda.Event()
                ev.record()
                defer_wgrad.append((ev, launch, (r["x"], dy, dwc, r["x_scale"], r["x_shift"])))
            else:
                with ops.SideStream(r["x"], dy, dwc):
                    launch(red=reduces)
        if li == 0 and not need_dx:
            break
        xin = r["x"]
        if r["stride"] == 1:
            if li > 0:               # g differentiates through the previous layer's BN + LeakyReLU: emit its partials here
                prev = sv["layers"][li - 1]
                g, part = ops.conv_dgrad_bwdstats(dy, wd[f"features.{ci}.weight"], w.shape[1], 3, prev["y"],
                                                  epi_scale=prev["scale"], epi_shift=prev["shift"], epi_slope_const=LRELU,
                                                  epi_act=1, grp=gB)
            else:
                g = ops.conv_fwd(dy, wd[f"features.{ci}.weight"], w.shape[1], 3, 1)[0]
        else:
            # the backward partials of g against the layer below come out of the data-gradient's epilogue where the pipelined kernel
            # takes the shape (no separate reduce pass over g and y); they are only needed when the layer below has a BatchNorm
            # or parameter gradients are wanted (its bias)
            prev = sv["layers"][li - 1] if li > 0 else None
            epi = None
            if prev is not None and (prev["bi"] is not None or wg):
                epi = dict(y=prev["y"], scale=prev["scale"], shift=prev["shift"], slope_const=LRELU, act=1)
            out = ops.conv_s2_dgrad(dy, ws2[f"features.{ci}.weight"], xin.shape[1], xin.shape[2], w.shape[1], epi=epi, grp=gB)
            g, part = out if epi is not None else (out, None)
        if li == 0:
            dx = ops.transpose(g, to_nchw=True)
    if reduces:
        # the weight gradients' slab reduces (leaves of the chain), all layers in one launch - on the stream the weight gradients ran on
        with ops.SideStream(*[t for j in reduces for t in j[:2]]):
            ops.wgrad_reduce_flush(reduces)
    ops.join_side()
    if acc:
        grads = {}                   # already added into the first pass's buffer
    return grads, dx


def backward(module, p, sv, dout, need_param_grads, need_dx):
    st = backward_classifier(module, p, sv, dout, need_param_grads)
    hook = module.__dict__.get("_after_cls_bwd")
    if hook is not None:             # engine.TrainEngine: the classifier's weights have been read for the last time on this stream
        hook()
    out = backward_features(module, p, sv, st, need_param_grads, need_dx)
    hook = module.__dict__.get("_after_bwd")
    if hook is not None:             # engine.TrainEngine: the generator's backward has left the discriminator
        hook()
    return out


class DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, grad_mode, *params):
        names = [n for n, _ in module.named_parameters()]
        p = dict(zip(names, [t.detach() for t in params]))
        need_param = grad_mode and any(ctx.needs_input_grad[3:])
        need_dx = grad_mode and ctx.needs_input_grad[0]
        need_grad = need_param or need_dx
        if need_grad and not module.training:
            raise NotImplementedError("Discriminator backward in eval() mode is not on the reference's path (train.py:110)")
        keep = module.__dict__.get("_keep_pass") and need_grad and module.training
        req = module.__dict__.get("_arena_request") if keep else None          # (slots, slot): write this pass into a fresh PassArena
        arena = (PassArena(req[0]), req[1]) if req is not None else None
        out, sv = forward(module, x, p, module.training, need_grad, arena=arena)
        if keep:
            # the step engine re-uses this pass (replay_running_stats): input identity, logits, saved activations
            module.__dict__["_last_pass"] = {"x_ptr": x.data_ptr(), "x_shape": tuple(x.shape), "out": out, "sv": sv, "p": p,
                                             "arena": arena[0] if arena is not None else None}
        if need_grad:
            ctx.module, ctx.sv, ctx.p, ctx.names = module, sv, p, names
            ctx.need_param, ctx.need_dx = need_param, need_dx
        return out

    @staticmethod
    def backward(ctx, dout):
        grads, dx = backward(ctx.module, ctx.p, ctx.sv, dout, ctx.need_param, ctx.need_dx)
        ctx.sv = None
        return (dx, None, None, *[grads.get(n) for n in ctx.names])

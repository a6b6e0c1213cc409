"""Step engines: the inner loops of reference warmup.py:74-96 and train.py:116-164 as objects.

    WarmupEngine.step(gt, lr)   G fwd -> sum_w criterion(sr,gt)*w -> bwd -> [RCCL all-reduce] -> Adam
    TrainEngine.step(gt, lr)    G update (D frozen but in train mode: its BN stats move, train.py:110,136),
                                then D update every D_UPDATE_INTERVAL-th batch on gt and sr.detach()

Same semantics as the reference step by step; what differs is execution:
  * the whole step (kernels of libsrganst.so + optimizer) is captured once into a hipGraph and
    replayed (no tracing compiler; the reference uses torch.compile, train.py:56);
  * loss values stay on the device; ``.item()`` happens only when the caller asks (the reference
    syncs per criterion per step, train.py:141);
  * data parallel: one process per GPU, gradients averaged with a single flat RCCL all-reduce
    (srganst.dist) between backward and the optimizer step.
"""
from __future__ import annotations

from collections import OrderedDict

import os

import torch

from . import dist as sdist
from . import ops as _ops


def make_adam(module, lr, betas, eps, weight_decay, capturable):
    # torch.optim.Adam semantics and state layout (the reference uses it as-is, train.py:62-75) on flat buffers: one
    # HIP kernel per step, graph-capturable, lr as a device tensor so that LR schedulers keep working under capture.
    from .optim import FlatAdam
    return FlatAdam(module, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable)


_ONES = {}


def _one(like):
    # persistent seed of backward(): autograd would otherwise launch a ones_like fill every step
    o = _ONES.get(like.device)
    if o is None:
        o = _ONES[like.device] = torch.ones((), device=like.device, dtype=torch.float32)
    return o


def _criterion_total(sr, gt, criterions, weights, adversarial=None, adv_logits=None, adv_label=None):
    """sum_name weight * criterion(sr, gt) as in the reference loops (train.py:129-140, warmup.py:79-86); the values come
    back per name (already weighted, detached).  HIP-path pixel / structure-tensor terms share one autograd node
    (loss.criterion_sum); "Adversarial" (through D) and any other criterion go through autograd term by term."""
    from .loss import criterion_sum, fusable
    vals = OrderedDict((name, None) for name in criterions)
    fused = [n for n, c in criterions.items() if n != "Adversarial" and fusable(c)]
    total = None
    if fused:
        total, weighted = criterion_sum(sr, gt, [criterions[n] for n in fused], [weights[n] for n in fused])
        for i, n in enumerate(fused):
            vals[n] = weighted[i]
    from .loss import BCEWithLogitsLoss, adversarial_term
    for name, crit in criterions.items():
        if name in fused:
            continue
        if name == "Adversarial" and adversarial is not None and adv_logits is not None and total is not None and isinstance(crit, BCEWithLogitsLoss):
            # total + w * BCE(D(sr), label) as ONE node: the scaling and the sum ride in the loss kernels (forward: one tiny launch
            # instead of mul + add; backward: the weight is folded into the BCE gradient's scale)
            total, vals[name] = adversarial_term(total, adv_logits(), crit, adv_label, weights[name])
            continue
        l = (adversarial(crit) if (name == "Adversarial" and adversarial is not None) else crit(sr, gt)) * weights[name]
        vals[name] = l.detach()
        total = l if total is None else total + l
    return total, vals


def _load_inputs(eng, gt, lr):
    """The captured steps read the engine's static input buffers eng.gt / eng.lr.  A batch handed in as other tensors is copied
    into them (device-to-device); a caller that fills the static buffers itself - `eng.gt.copy_(host_batch, non_blocking=True)`
    returns eng.gt - and passes them back pays no extra copy."""
    if eng.gt is None:
        eng.gt, eng.lr = gt.clone(), lr.clone()
        return
    if gt is not eng.gt:
        eng.gt.copy_(gt, non_blocking=True)
    if lr is not eng.lr:
        eng.lr.copy_(lr, non_blocking=True)


class _GraphedStep:
    """Capture ``fn()`` (which reads the static input buffers) into a hipGraph after a few eager warm-up calls."""

    def __init__(self, fn, warmup_calls=2, enabled=True, on_fail=None):
        self.fn = fn
        self.graph = None
        self.enabled = enabled
        self.calls = 0
        self.warmup_calls = warmup_calls
        self.out = None
        self.on_fail = on_fail      # called once when a capture fails: the owner drops EVERY graph of the engine (see _drop_graphs)

    def drop(self):
        """Back to eager for good (a sibling step's capture failed: graphs bake in each other's static tensors)."""
        self.graph = None
        self.enabled = False

    def __call__(self):
        if not self.enabled:
            return self.fn()
        if self.graph is None:
            if self.calls < self.warmup_calls:
                self.calls += 1
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    out = self.fn()
                torch.cuda.current_stream().wait_stream(s)
                return out
            launch_stream = torch.cuda.current_stream()
            # No cyclic garbage collection while the stream captures: a collection that happens to free an OLD CUDAGraph (a
            # previous engine of the process) calls hipGraphDestroy inside the open capture - "operation not permitted when stream
            # is capturing", thrown from the destructor, process terminated (seen in tests that build several engines in a row).
            import gc
            gc_was_enabled = gc.isenabled()
            gc.disable()
            try:
                g = torch.cuda.CUDAGraph()
                dot = os.environ.get("SST_GRAPH_DOT")       # dev: directory for hipGraphDebugDotPrint dumps of every captured graph
                if dot:
                    g.enable_debug_mode()
                # thread_local: calls made by OTHER threads during capture (e.g. the RCCL watchdog polling its events)
                # must not invalidate it
                # a capture stream of its own: torch's shared default capture stream would stay dead for every later capture
                # of the process once ONE capture on it has been invalidated
                cap_stream = torch.cuda.Stream()
                _ops.TOPOLOGY_ERROR = None
                _ops.CAPTURE_ORIGIN = cap_stream          # helper-stream joins are checked against it (ops.check_capture_join)
                try:
                    with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local"):
                        self.out = self.fn()
                finally:
                    _ops.CAPTURE_ORIGIN = None
                self.graph = g
                if dot:
                    g.debug_dump(os.path.join(dot, f"graph_{getattr(self.fn, '__name__', 'fn')}_{id(self):x}.dot"))
            except Exception as e:  # keep training (eager) rather than die: a step is still the same kernels
                import sys
                if _ops.TOPOLOGY_ERROR is not None:      # a schedule the runtime cannot capture is a configuration error, not a fallback case
                    msg, _ops.TOPOLOGY_ERROR = _ops.TOPOLOGY_ERROR, None
                    torch.cuda.set_stream(launch_stream)
                    torch.cuda.synchronize()
                    from . import _abi as _a
                    _a.lib().sst_clear_error()
                    self.drop()
                    if self.on_fail is not None:
                        self.on_fail()
                    raise _ops.CaptureTopologyError(msg) from e
                print(f"[srganst] hipGraph capture failed ({type(e).__name__}: {e}); the whole engine continues in eager mode",
                      file=sys.stderr)
                # torch.cuda.graph.__exit__ raises from capture_end() BEFORE it leaves its stream context: the current stream
                # would stay the dead (invalidated) capture stream, on which every later launch fails
                torch.cuda.set_stream(launch_stream)
                g = None
                torch.cuda.synchronize()
                from . import _abi
                _abi.lib().sst_clear_error()       # the failed capture leaves a sticky runtime error: the next launch check would trip on it
                self.drop()
                if self.on_fail is not None:
                    self.on_fail()
                if gc_was_enabled:
                    gc.enable()
                return self.fn()
            finally:
                if gc_was_enabled:
                    gc.enable()
        self.graph.replay()
        return self.out


class WarmupEngine:
    """reference warmup.py:14-96 without the data loader / logging / validation around the step."""

    def __init__(self, config, generator, criterions=None, weights=None, use_graph=None, process_group=None,
                 adam_capturable=None, force_dp=False):
        self.config = config
        self.G = generator
        self.criterions = criterions if criterions is not None else config.MODEL.G_LOSS.WARMUP_CRITERIONS
        self.weights = weights if weights is not None else config.MODEL.G_LOSS.WARMUP_WEIGHTS
        self.pg = process_group
        self.world = sdist.world_size(process_group)
        use_graph = config.KERNEL.USE_GRAPH if use_graph is None else use_graph
        self.opt = make_adam(self.G, config.SOLVER.G_BASE_LR, (config.SOLVER.G_BETA1, config.SOLVER.G_BETA2),
                             config.SOLVER.G_EPS, config.SOLVER.G_WEIGHT_DECAY,
                             capturable=use_graph if adam_capturable is None else adam_capturable)
        self.gt = self.lr = None
        self.loss_values = OrderedDict()
        self.sr = None
        self.dp = self.world > 1 or force_dp     # force_dp: run the split-graph + collective path even with one rank (tests)
        # RCCL: the all-reduce is captured inside the step's graph (DIST.ONE_GRAPH); otherwise it stays outside:
        # [fwd+bwd graph] -> all-reduce -> [optimizer]
        self.one_graph_dp = self.dp and sdist.is_rccl(process_group) and bool(config.DIST.ONE_GRAPH)
        if self.one_graph_dp:
            self._fb = _GraphedStep(self._full_step_dp, enabled=use_graph, on_fail=self._drop_graphs)
            self._op = None
        elif self.dp:
            self._fb = _GraphedStep(self._fwd_bwd, enabled=use_graph, on_fail=self._drop_graphs)
            self._op = _GraphedStep(self._opt_step, enabled=use_graph, on_fail=self._drop_graphs)
        else:
            self._fb = _GraphedStep(self._full_step, enabled=use_graph, on_fail=self._drop_graphs)
            self._op = None

    def _steps(self):
        return [s for s in (self._fb, self._op) if s is not None]

    def _drop_graphs(self):
        for s in self._steps():
            s.drop()

    @property
    def graph_active(self):
        """True when every part of the step is replayed from a captured hipGraph (False before capture and after a fallback)."""
        st = self._steps()
        return bool(st) and all(s.graph is not None for s in st)

    # -- pieces
    def _fwd_bwd(self):
        self.opt.zero_grad(set_to_none=True)
        sr = self.G(self.lr)
        total, vals = _criterion_total(sr, self.gt, self.criterions, self.weights)
        total.backward(_one(total))
        self.sr = sr.detach()
        self.loss_values = vals
        return vals

    def _opt_step(self):
        self.opt.step()

    def _full_step(self):
        vals = self._fwd_bwd()
        self.opt.step()
        return vals

    def _full_step_dp(self):
        vals = self._fwd_bwd()
        ar = sdist.AsyncAllReduce(sdist.module_flat_grad(self.G), self.pg, force=True)      # forks from / joins this (the origin) stream
        if ar.flat is None:
            sdist.allreduce_module_grads(self.G, self.pg, force=True)
        ar.wait()
        self.opt.step()
        return vals

    def close(self):
        """Drop the captured graphs and static buffers (also breaks the engine <-> bound-method reference cycle, so the
        device objects are released right here and not by a later cyclic collection)."""
        self._fb = self._op = None
        self.gt = self.lr = self.sr = None

    def step(self, gt, lr):
        """One optimisation step on the batch (gt [B,3,H,W], lr [B,3,H/4,W/4], device tensors)."""
        _load_inputs(self, gt, lr)
        self._fb()
        if self.dp and not self.one_graph_dp:
            sdist.allreduce_module_grads(self.G, self.pg, force=True)
            self._op()
        return self.loss_values


class TrainEngine:
    """reference train.py:16-164 without loaders / logging / validation around the step."""

    def __init__(self, config, generator, discriminator, use_graph=None, process_group=None, adam_capturable=None,
                 force_dp=False, overlap_comm=None):
        from .loss import BCEWithLogitsLoss
        self.config = config
        self.G, self.D = generator, discriminator
        self.pg = process_group
        self.world = sdist.world_size(process_group)
        use_graph = config.KERNEL.USE_GRAPH if use_graph is None else use_graph
        s = config.SOLVER
        cap = use_graph if adam_capturable is None else adam_capturable
        self.g_opt = make_adam(self.G, s.G_BASE_LR, (s.G_BETA1, s.G_BETA2), s.G_EPS, s.G_WEIGHT_DECAY, cap)
        self.d_opt = make_adam(self.D, s.D_BASE_LR, (s.D_BETA1, s.D_BETA2), s.D_EPS, s.D_WEIGHT_DECAY, cap)
        self.adv = BCEWithLogitsLoss()                      # train.py:59
        self.real = 1.0 - config.EXP.LABEL_SMOOTHING        # train.py:113
        self.fake = 0.0                                     # train.py:114
        self.batch_num = 0
        self.gt = self.lr = self.sr = None
        self.loss_values = OrderedDict()
        self.d_loss = self.pred_gt = self.pred_sr = None
        g = use_graph
        self.dp = self.world > 1 or force_dp     # force_dp: the split-graph + collective path even with one rank (tests)
        # D's packed weights are made once per iteration, by the D(sr) of the generator step, and re-used by the forward and
        # backward passes of the discriminator step (disc_graph._packs): this engine owns every update of D's weights
        self.D.__dict__["_packs_managed"] = True
        self.D.__dict__["_packs_fresh"] = False
        # A captured graph bakes in the tensors that exist at capture time (D's graph reads the generator graph's static `sr`):
        # a mixed eager / graph state would replay on stale buffers, so ONE failed capture sends the whole engine back to eager.
        f = self._drop_graphs
        # Data parallel with overlap (default): the discriminator half runs without autograd as [forward of both passes +
        # classifier backward] -> classifier bucket on the wire -> [feature-stack backward] -> feature bucket; the
        # generator's message travels under the first of those graphs.  overlap_comm=False: the round-1 schedule
        # ([fwd+bwd] -> all-reduce -> wait -> [Adam] per network); both give bit-identical parameters (tests/test_dp_gpu.py).
        self.overlap = self.dp and (config.DIST.OVERLAP_COMM if overlap_comm is None else overlap_comm)
        # ... and with RCCL the collectives are captured inside the merged iteration graph (_iter_gd): ONE graph per iteration at N > 1
        # as at N = 1.  (What hipStreamEndCapture accepts decides the shape: a collective issued on the side branch runs on the process
        # group's stream, a SECOND-level fork, which may only be joined into the graph's origin stream - see _iter_gd.)
        self.one_graph_dp = (self.overlap and sdist.is_rccl(process_group) and bool(config.DIST.ONE_GRAPH)
                             and bool(getattr(config.KERNEL, "OVERLAP_GD", False)))
        if self.one_graph_dp:
            self.overlap = False
        # The generator's gradient travels on a communicator of its own in the overlapped schedule: on the default one its 6.2 MB
        # message would queue behind the discriminator's buckets (collectives of one process group run in issue order on one stream)
        # and hold the generator's Adam until the feature bucket - i.e. the whole discriminator backward - is through.
        self.pg_g = process_group
        if self.overlap and self.world > 1 and sdist.is_rccl(process_group) and bool(config.DIST.G_OWN_GROUP):
            import torch.distributed as td
            self.pg_g = td.new_group(ranks=list(range(td.get_world_size())) if process_group is None else td.get_process_group_ranks(process_group))
        self._d_a = self._d_b = self._g_f = self._g_b = None
        self.d_batched = False           # set once the discriminator step has run its two passes as one batch (KERNEL.BATCH_D_STEP)
        self.d_sr_reused = False         # set once the discriminator step has run on the generator step's D(sr) pass (KERNEL.REUSE_D_SR)
        self._side = self._side_d = None
        self._it = None
        if self.one_graph_dp:
            self._g_fb, self._g_op = _GraphedStep(self._g_full, enabled=g, on_fail=f), None
            self._d_fb = self._d_op = None
            self._it = _GraphedStep(self._iter_gd, enabled=g, on_fail=f)
        elif self.dp:
            self._g_fb, self._g_op = _GraphedStep(self._g_fwd_bwd, enabled=g, on_fail=f), _GraphedStep(self.g_opt.step, enabled=g, on_fail=f)
            self._d_fb, self._d_op = _GraphedStep(self._d_fwd_bwd, enabled=g, on_fail=f), _GraphedStep(self._d_step, enabled=g, on_fail=f)
            if self.overlap:
                self._d_fb = self._g_fb = None
                self._g_f = _GraphedStep(self._g_fwd, enabled=g, on_fail=f)         # generator half split at the end of its forward:
                self._g_b = _GraphedStep(self._g_bwd, enabled=g, on_fail=f)         # the discriminator branch starts there
                self._d_a = _GraphedStep(self._d_fwd_cls, enabled=g, on_fail=f)
                self._d_b = _GraphedStep(self._d_features, enabled=g, on_fail=f)
        else:
            two = bool(getattr(config.KERNEL, "D_TWO_STREAMS", False))
            self._g_fb, self._g_op = _GraphedStep(self._g_full, enabled=g, on_fail=f), None
            self._d_fb, self._d_op = _GraphedStep(self._d_two_stream_full if two else self._d_full, enabled=g, on_fail=f), None
            # whole iteration as ONE graph: the discriminator step runs beside the generator's backward (see _iter_gd)
            if bool(getattr(config.KERNEL, "OVERLAP_GD", False)):
                self._it = _GraphedStep(self._iter_gd, enabled=g, on_fail=f)

    def _steps(self):
        if self._it is not None:               # merged iterations; generator-only graph between them when D is updated every n-th step
            return [self._it] + ([self._g_fb] if self.config.SOLVER.D_UPDATE_INTERVAL > 1 else [])
        return [s for s in (self._g_fb, self._g_f, self._g_b, self._g_op, self._d_fb, self._d_a, self._d_b, self._d_op) if s is not None]

    def _drop_graphs(self):
        for s in self._steps():
            s.drop()

    @property
    def graph_active(self):
        """True when every part of the iteration is replayed from a captured hipGraph."""
        st = self._steps()
        return bool(st) and all(s.graph is not None for s in st)

    # -- generator half: train.py:125-144
    def _g_fwd_bwd(self):
        cfg = self.config
        for p in self.D.parameters():
            p.requires_grad = False
        self.g_opt.zero_grad(set_to_none=True)
        self.D.__dict__["_packs_fresh"] = False      # the generator step always packs D's current weights (also when captured)
        sr = self.G(self.lr)
        total, vals = _criterion_total(sr, self.gt, cfg.MODEL.G_LOSS.CRITERIONS, cfg.MODEL.G_LOSS.CRITERION_WEIGHTS,
                                       adversarial=lambda crit: crit(self.D(sr), self.real),
                                       adv_logits=lambda: self.D(sr), adv_label=self.real)
        total.backward(_one(total))
        self.sr = sr.detach()
        self.loss_values = vals
        return vals

    # the same half in two parts (data-parallel schedule: the discriminator branch forks between them)
    def _g_fwd(self):
        cfg = self.config
        for p in self.D.parameters():
            p.requires_grad = False
        self.g_opt.zero_grad(set_to_none=True)
        self.D.__dict__["_packs_fresh"] = False
        self.D.__dict__["_keep_pass"], self.D.__dict__["_last_pass"] = True, None      # _d_fwd_cls re-uses the D(sr) pass
        self._request_arena()
        early_pack = cfg.KERNEL.EARLY_D_PACK and "Adversarial" in cfg.MODEL.G_LOSS.CRITERIONS
        if early_pack:                             # D's weight packing beside the generator's forward (as in _iter_gd)
            from . import disc_graph
            main = torch.cuda.current_stream()
            if self._side_d is None:
                self._side_d = torch.cuda.Stream()
            names = [n for n, _ in self.D.named_parameters()]
            self._side_d.wait_stream(main)
            with torch.cuda.stream(self._side_d):
                disc_graph._packs(self.D, dict(zip(names, [t.detach() for t in self.D.parameters()])), True)
        sr = self.G(self.lr)
        if early_pack:
            main.wait_stream(self._side_d)
        total, vals = _criterion_total(sr, self.gt, cfg.MODEL.G_LOSS.CRITERIONS, cfg.MODEL.G_LOSS.CRITERION_WEIGHTS,
                                       adversarial=lambda crit: crit(self.D(sr), self.real),
                                       adv_logits=lambda: self.D(sr), adv_label=self.real)
        self.D.__dict__["_keep_pass"] = False
        self.D.__dict__.pop("_arena_request", None)
        self.sr = sr.detach()
        self.loss_values = vals
        self._g_total = total
        return vals

    def _g_bwd(self):
        self._g_total.backward(_one(self._g_total))

    def _g_allreduce(self):
        """Mean of the generator's gradient over the ranks, issued from (and joined into) the current stream; capturable with RCCL."""
        ar = sdist.AsyncAllReduce(sdist.module_flat_grad(self.G), self.pg, force=True)
        if ar.flat is None:                                      # gradients not in one flat buffer: the generic path
            sdist.allreduce_module_grads(self.G, self.pg, force=True)
        ar.wait()

    def _g_full(self):
        v = self._g_fwd_bwd()
        if self.one_graph_dp:
            self._g_allreduce()
        self.g_opt.step()
        return v

    def _request_arena(self):
        """KERNEL.REUSE_D_SR + KERNEL.BATCH_D_STEP: the generator step's D(sr) pass writes its activations into slot 1 of a two-pass arena
        (disc_graph.PassArena); the discriminator step's D(gt) fills slot 0 and ONE backward runs over both (_d_fwd_cls)."""
        cfg = self.config
        ok = (cfg.KERNEL.REUSE_D_SR and cfg.KERNEL.BATCH_D_STEP and "Adversarial" in cfg.MODEL.G_LOSS.CRITERIONS
              and self.batch_num % cfg.SOLVER.D_UPDATE_INTERVAL == 0 and self.gt is not None)
        if ok:
            from . import disc_graph
            names = [n for n, _ in self.D.named_parameters()]
            pd = dict(zip(names, [t.detach() for t in self.D.parameters()]))
            ok = disc_graph.groups_supported(self.D, pd, self.gt.shape[0], 2, self.gt.shape[2], self.gt.shape[3])
        if ok:
            self.D.__dict__["_arena_request"] = (2, 1)
        else:
            self.D.__dict__.pop("_arena_request", None)

    # -- discriminator half: train.py:149-164
    def _d_fwd_bwd(self):
        for p in self.D.parameters():
            p.requires_grad = True
        self.d_opt.zero_grad(set_to_none=True)
        pred_gt = self.D(self.gt)
        loss_real = self.adv(pred_gt, self.real)
        pred_sr = self.D(self.sr)                            # train.py:158 detaches + clones; self.sr is detached and D only reads it
        loss_fake = self.adv(pred_sr, self.fake)
        d_loss = loss_real + loss_fake
        scope = {"flat": None}
        self.D.__dict__["_grad_accum"] = scope               # both backward passes write ONE flat gradient buffer (disc_graph.backward)
        try:
            d_loss.backward(_one(d_loss))
        finally:
            self.D.__dict__.pop("_grad_accum", None)
        # The scope relies on autograd ADOPTING the first pass's views as p.grad (no clone) while the second pass adds into the
        # same buffer and returns nothing.  If a torch version / a hook ever clones instead, the second pass's half of the
        # gradient would be lost silently: check the aliasing once per (eager or capturing) call.
        flat = scope["flat"]
        if flat is not None:
            lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
            for n, p in self.D.named_parameters():
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError(f"discriminator gradient of {n} is not a view of the accumulation buffer: "
                                       "the second backward pass would be dropped (engine._d_fwd_bwd)")
        self.d_loss, self.pred_gt, self.pred_sr = d_loss.detach(), pred_gt.detach(), pred_sr.detach()
        return self.d_loss

    # -- discriminator half without autograd, in two parts (data-parallel overlap).  Same kernels with the same arguments as
    # the autograd path above, in an order that keeps every parameter's accumulation order (autograd runs the D(sr) pass
    # before the D(gt) pass: loss_fake was recorded last), so the gradients are bit-identical.
    def _d_gt_fwd(self):
        """D(gt)'s forward ahead of its place in the reference's sequence (it needs nothing of the generator): running statistics
        and batch counter untouched - _d_fwd_cls replays them where the reference has the pass (after the generator step's D(sr))."""
        from . import disc_graph
        D = self.D
        names = [n for n, _ in D.named_parameters()]
        pd = dict(zip(names, [t.detach() for t in D.parameters()]))
        pred_gt, sv_gt = disc_graph.forward(D, self.gt, pd, True, True, update_running=False)
        return pd, pred_gt, sv_gt

    def _d_fwd_cls(self, early_gt=None):
        from . import disc_graph, ops
        D = self.D
        for p in D.parameters():
            p.requires_grad = True
        self.d_opt.zero_grad(set_to_none=True)
        names = [n for n, _ in D.named_parameters()]
        kept = D.__dict__.get("_last_pass") if self.config.KERNEL.REUSE_D_SR else None
        if early_gt is None and kept is None and self.config.KERNEL.BATCH_D_STEP:
            pd = dict(zip(names, [t.detach() for t in D.parameters()]))
            B = self.gt.shape[0]
            if self.sr.shape == self.gt.shape and disc_graph.groups_supported(D, pd, B, 2, self.gt.shape[2], self.gt.shape[3]):
                # Both passes of the step as ONE batch [gt ; sr] with per-pass BatchNorm statistics (running statistics move in the
                # reference's order: gt, then sr), one backward over 2B images.
                D.__dict__.pop("_last_pass", None)
                pred, sv = disc_graph.forward(D, [self.gt, self.sr], pd, True, True)
                pred_gt, pred_sr = pred[:B], pred[B:]
                dl = torch.empty_like(pred)
                loss_real, _ = ops.bce_logits(pred_gt, self.real, want_loss=True, want_grad=True, grad_out=dl[:B])
                loss_fake, _ = ops.bce_logits(pred_sr, self.fake, want_loss=True, want_grad=True, grad_out=dl[B:])
                self.d_loss, self.pred_gt, self.pred_sr = loss_real + loss_fake, pred_gt, pred_sr
                st = disc_graph.backward_classifier(D, pd, sv, dl, True)
                self._d_state = (pd, None, None, sv, st)
                self.d_batched = True
                flat = D.__dict__["_flat_grads"][-1]
                plist = [pd[n] for n in names]
                offs, total = ops.flat_layout(plist)
                cut = offs[names.index("classifier.0.weight")]
                self._d_flat, self._d_buckets = flat, (flat[cut:total], flat[:cut])
                return self.d_loss
        if early_gt is None and kept is not None and kept.get("arena") is not None and self.config.KERNEL.BATCH_D_STEP:
            pd = dict(zip(names, [t.detach() for t in D.parameters()]))
            if (kept["x_ptr"] == self.sr.data_ptr() and kept["x_shape"] == tuple(self.sr.shape) and tuple(self.gt.shape) == tuple(self.sr.shape)
                    and all(kept["p"][n].data_ptr() == pd[n].data_ptr() and kept["p"][n]._version == pd[n]._version for n in names)):
                # The kept D(sr) pass sits in slot 1 of a two-pass arena: D(gt) fills slot 0, the running statistics take D(sr.detach())'s
                # step (replayed, as below), and ONE backward runs over the 2B images with per-pass BatchNorm rows.
                D.__dict__.pop("_last_pass", None)
                B = self.gt.shape[0]
                arena = kept["arena"]
                pred_gt, sv_gt = disc_graph.forward(D, self.gt, pd, True, True, arena=(arena, 0))
                disc_graph.replay_running_stats(D, pd, kept["sv"])
                pred_sr = kept["out"].detach()
                self.d_sr_reused = True
                dl = torch.empty(2 * B, 1, device=pred_gt.device, dtype=torch.float32)
                loss_real, _ = ops.bce_logits(pred_gt, self.real, want_loss=True, want_grad=True, grad_out=dl[:B])
                loss_fake, _ = ops.bce_logits(pred_sr, self.fake, want_loss=True, want_grad=True, grad_out=dl[B:])
                self.d_loss, self.pred_gt, self.pred_sr = loss_real + loss_fake, pred_gt, pred_sr
                sv = disc_graph.batched_saved(arena, sv_gt)
                st = disc_graph.backward_classifier(D, pd, sv, dl, True)
                self._d_state = (pd, None, None, sv, st)
                self.d_batched = True
                flat = D.__dict__["_flat_grads"][-1]
                plist = [pd[n] for n in names]
                offs, total = ops.flat_layout(plist)
                cut = offs[names.index("classifier.0.weight")]
                self._d_flat, self._d_buckets = flat, (flat[cut:total], flat[:cut])
                return self.d_loss
        if early_gt is not None:
            pd, pred_gt, sv_gt = early_gt
            disc_graph.replay_running_stats(D, pd, sv_gt)
        else:
            pd = dict(zip(names, [t.detach() for t in D.parameters()]))
            pred_gt, sv_gt = disc_graph.forward(D, self.gt, pd, True, True)
        loss_real, dl_gt = ops.bce_logits(pred_gt, self.real, want_loss=True, want_grad=True)
        # D(sr.detach()) (train.py:158) repeats the generator step's D(sr) (train.py:136): same input, same weights (D's Adam comes
        # after both), train-mode BatchNorm both times - every kernel is deterministic, so the pass would reproduce the saved
        # activations and logits bit for bit.  It is not run again: its side effects (running statistics, batch counter) are
        # replayed in the reference's order (after D(gt)'s) and the backward works on the generator step's saved pass.
        kept = D.__dict__.pop("_last_pass", None) if self.config.KERNEL.REUSE_D_SR else None
        if (kept is not None and kept["x_ptr"] == self.sr.data_ptr() and kept["x_shape"] == tuple(self.sr.shape)
                and all(kept["p"][n].data_ptr() == pd[n].data_ptr() and kept["p"][n]._version == pd[n]._version for n in names)):
            disc_graph.replay_running_stats(D, pd, kept["sv"])
            pred_sr, sv_sr = kept["out"].detach(), kept["sv"]
            self.d_sr_reused = True
        else:
            pred_sr, sv_sr = disc_graph.forward(D, self.sr, pd, True, True)
        loss_fake, dl_sr = ops.bce_logits(pred_sr, self.fake, want_loss=True, want_grad=True)
        self.d_loss, self.pred_gt, self.pred_sr = loss_real + loss_fake, pred_gt, pred_sr
        D.__dict__["_grad_accum"] = {"flat": None}
        try:
            st_sr = disc_graph.backward_classifier(D, pd, sv_sr, dl_sr, True)
            st_gt = disc_graph.backward_classifier(D, pd, sv_gt, dl_gt, True)
        finally:
            D.__dict__.pop("_grad_accum", None)
        self._d_state = (pd, sv_sr, st_sr, sv_gt, st_gt)
        flat = D.__dict__["_flat_grads"][-1]
        plist = [pd[n] for n in names]
        offs, total = ops.flat_layout(plist)
        cut = offs[names.index("classifier.0.weight")]          # features.* come first in the reference's parameter order
        self._d_flat, self._d_buckets = flat, (flat[cut:total], flat[:cut])
        return self.d_loss

    def _d_features(self, defer_gt_wgrad=None):
        """defer_gt_wgrad (a list): the conv weight gradients of the D(gt) pass are handed back instead of launched
        (disc_graph.backward_features); they ADD into what the D(sr) pass wrote, which is complete by then on this stream."""
        from . import disc_graph
        pd, sv_sr, st_sr, sv_gt, st_gt = self._d_state
        if sv_sr is None:               # both passes as one batch (_d_fwd_cls): one backward, its weight gradients are the leaves
            grads, _ = disc_graph.backward_features(self.D, pd, sv_gt, st_gt, True, False, defer_wgrad=defer_gt_wgrad,
                                                    defer_below=int(self.config.KERNEL.DEFER_D_WGRAD_BATCHED))
        else:
            grads, _ = disc_graph.backward_features(self.D, pd, sv_sr, st_sr, True, False)
            disc_graph.backward_features(self.D, pd, sv_gt, st_gt, True, False, defer_wgrad=defer_gt_wgrad,
                                         defer_below=int(self.config.KERNEL.DEFER_D_WGRAD))
        for n, p in self.D.named_parameters():
            p.grad = grads[n]

    # -- discriminator half with its two passes on two streams (under capture: two parallel branches of the hipGraph).  D(gt) and
    # D(sr.detach()) share nothing but the weights (read-only here), the BatchNorm running statistics (ordered pass by pass at
    # every bn_finalize, so the buffers move exactly as in the sequential order gt -> sr) and the gradient buffer (each pass
    # writes its own flat buffer; they are added afterwards: fl(a + b), bit for bit what the accumulate flag does).  The many
    # latency-bound launches of one pass (BatchNorm finalize / backward reduce / apply, 1-4 workgroups each) then run beside
    # the other pass's kernels instead of in front of them.  Measured: 6.49 -> 6.18 ms per iteration; superseded by _iter_gd
    # (5.92 ms), kept as KERNEL.D_TWO_STREAMS for engines that do not merge the iteration.
    def _d_two_stream(self):
        from . import disc_graph, ops
        D = self.D
        for p in D.parameters():
            p.requires_grad = True
        self.d_opt.zero_grad(set_to_none=True)
        names = [n for n, _ in D.named_parameters()]
        pd = dict(zip(names, [t.detach() for t in D.parameters()]))
        ops.flatten_bn_counters(D).add_(2)
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        ev = {}

        def hook_a(li, when):
            if when == "post":
                e = torch.cuda.Event()
                e.record(main)
                ev[li] = e

        def hook_b(li, when):
            if when == "pre":
                side.wait_event(ev[li])

        pred_gt, sv_gt = disc_graph.forward(D, self.gt, pd, True, True, bump_counters=False, bn_hook=hook_a)   # packs weights if needed
        loss_real, dl_gt = ops.bce_logits(pred_gt, self.real, want_loss=True, want_grad=True)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pred_sr, sv_sr = disc_graph.forward(D, self.sr, pd, True, True, bump_counters=False, bn_hook=hook_b)
            loss_fake, dl_sr = ops.bce_logits(pred_sr, self.fake, want_loss=True, want_grad=True)
            st_sr = disc_graph.backward_classifier(D, pd, sv_sr, dl_sr, True)
            g_sr, _ = disc_graph.backward_features(D, pd, sv_sr, st_sr, True, False)
            flat_sr = D.__dict__["_flat_grads"][-1]
        st_gt = disc_graph.backward_classifier(D, pd, sv_gt, dl_gt, True)
        g_gt, _ = disc_graph.backward_features(D, pd, sv_gt, st_gt, True, False)
        flat_gt = D.__dict__["_flat_grads"][-1]
        ops.check_capture_join(main)          # inside the side branch of a merged capture this join is the one the runtime faults on
        main.wait_stream(side)
        flat_sr.add_(flat_gt)                 # autograd order of the sequential path: the D(sr) pass writes, the D(gt) pass accumulates
        self.d_loss, self.pred_gt, self.pred_sr = loss_real + loss_fake, pred_gt, pred_sr
        lst = D.__dict__["_flat_grads"]
        lst[:] = [t for t in lst if t is not flat_sr] + [flat_sr]      # the buffer that holds p.grad is the newest one (FlatAdam / dist look there first)
        for n, p in D.named_parameters():
            p.grad = g_sr[n]
        return self.d_loss

    # -- the whole iteration (train.py:125-164) as one launch DAG: the discriminator step needs nothing of the generator's
    # backward (only sr, D's weights and - for the order of the running statistics - the generator step's D(sr) FORWARD), so it
    # starts as soon as the generator step's forward is complete and runs on side streams beside the generator's backward +
    # Adam: the trunk's backward is a chain of ~70 short launches that leave most of the chip idle.  D's Adam comes after the
    # join (the generator's backward through D still reads D's weights).  Same kernels, same arguments, same order per tensor:
    # bit-identical to the sequential schedule.
    def _iter_gd(self):
        try:
            return self._iter_gd_body()
        finally:
            self.D.__dict__["_counters_external"] = False

    def _iter_gd_body(self):
        cfg = self.config
        for p in self.D.parameters():
            p.requires_grad = False
        self.g_opt.zero_grad(set_to_none=True)
        self.D.__dict__["_packs_fresh"] = False
        self.D.__dict__["_keep_pass"], self.D.__dict__["_last_pass"] = True, None
        self._request_arena()
        from . import disc_graph, ops
        adv_d = "Adversarial" in cfg.MODEL.G_LOSS.CRITERIONS
        # the batch counters of D's BatchNorms move by one per pass (run or replayed): ONE add per iteration instead of three
        self.D.__dict__["_counters_external"] = True
        n_pass = 3 if adv_d else 2
        adv = adv_d
        ride = bool(adv and cfg.KERNEL.EARLY_D_PACK)       # ... and that add rides in the early pack launch below when there is one
        if not ride:
            ops.flatten_bn_counters(self.D).add_(n_pass)
        ops.debug_stamp(0)
        main = torch.cuda.current_stream()
        if self._side_d is None:
            self._side_d = torch.cuda.Stream()
        early_gt = None
        if adv and (cfg.KERNEL.EARLY_D_PACK or cfg.KERNEL.EARLY_D_GT):
            # D's weights are packed (one multi-tensor launch + one per stride-2 layer, 67 us) for all passes of the iteration on the
            # side stream, beside the generator's forward, instead of in front of D(sr) on the critical path
            names = [n for n, _ in self.D.named_parameters()]
            self._side_d.wait_stream(main)
            with torch.cuda.stream(self._side_d):
                disc_graph._packs(self.D, dict(zip(names, [t.detach() for t in self.D.parameters()])), True,
                                  counter_add=n_pass if ride else 0)
                if cfg.KERNEL.EARLY_D_GT:            # D(gt)'s forward beside the generator's forward as well (measured slower, off)
                    early_gt = self._d_gt_fwd()
        sr = self.G(self.lr)
        if adv and (cfg.KERNEL.EARLY_D_PACK or cfg.KERNEL.EARLY_D_GT):
            main.wait_stream(self._side_d)           # D(sr) below reads the packed weights
        ops.debug_stamp(1)
        total, vals = _criterion_total(sr, self.gt, cfg.MODEL.G_LOSS.CRITERIONS, cfg.MODEL.G_LOSS.CRITERION_WEIGHTS,
                                       adversarial=lambda crit: crit(self.D(sr), self.real),
                                       adv_logits=lambda: self.D(sr), adv_label=self.real)
        self.D.__dict__["_keep_pass"] = False
        self.D.__dict__.pop("_arena_request", None)
        self.sr = sr.detach()
        ops.debug_stamp(2)
        side = self._side_d
        dp_ar = {}

        def start_side():                       # the discriminator step's branch forks HERE off the stream the caller runs on (main)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # both passes on ONE side stream (a helper stream forked from it could only be joined into
                ops.debug_stamp(3)              # `main`: ops.check_capture_join)
                self._d_fwd_cls(early_gt)
                ops.debug_stamp(4)
                # data parallel, one graph: the classifier bucket (75.5 MB) goes out now, under the feature stack's backward.  The
                # collective runs on the process group's own stream - forked from this side stream, i.e. a second-level fork: its
                # wait() below is issued on `main`, the capture's origin stream (a second-level stream joined into a first-level one
                # is what crashes hipStreamEndCapture).
                dp_ar["c"] = sdist.AsyncAllReduce(self._d_buckets[0], self.pg, force=True) if self.one_graph_dp else None
        fork_at = int(cfg.KERNEL.FORK_D_STEP_AT) if adv_d else 0
        if fork_at == 0:
            start_side()
        else:
            self.D.__dict__["_after_cls_bwd" if fork_at == 1 else "_after_bwd"] = start_side
        # D's Adam in two parts (KERNEL.SPLIT_D_ADAM): the classifier's gradient is complete here; its weights are read once more,
        # by the head of the generator's backward (through D), which hands over with an event - then the classifier's update runs on
        # the side stream beside the rest of both branches and only the feature stack's (4.7 M parameters) is left for the join.
        names = [n for n, _ in self.D.named_parameters()]
        cls0 = names.index("classifier.0.weight")
        split = (bool(cfg.KERNEL.SPLIT_D_ADAM) and not self.dp and fork_at == 0 and self._d_flat is not None
                 and hasattr(self.d_opt, "step_params"))

        def cls_adam():
            ev = torch.cuda.Event()
            ev.record()                          # on the stream the backward runs on
            with torch.cuda.stream(side):
                side.wait_event(ev)
                self.d_opt.step_params(cls0, len(names), flat_grad=self._d_flat)
        if split:
            if adv_d:
                self.D.__dict__["_after_cls_bwd"] = cls_adam
            else:
                cls_adam()
        ops.debug_stamp(6)
        # The generator's weight gradients selected by KERNEL.DEFER_G_WGRAD leave its backward chain and run on the side stream behind
        # the discriminator step's work (gen_graph.backward hands them over): its Adam then waits for the join.
        g_def = [] if (int(cfg.KERNEL.DEFER_G_WGRAD) and not self.dp) else None
        if g_def is not None:
            self.G.__dict__["_defer_wgrad"] = (g_def, int(cfg.KERNEL.DEFER_G_WGRAD))
        try:
            with torch.autograd.set_multithreading_enabled(False):      # backward on this thread: one thread feeds the open capture
                total.backward(_one(total))
        finally:
            self.D.__dict__.pop("_after_cls_bwd", None)
            self.D.__dict__.pop("_after_bwd", None)
            self.G.__dict__.pop("_defer_wgrad", None)
        ops.debug_stamp(7)
        self.loss_values = vals
        if self.one_graph_dp:
            self._g_allreduce()                 # 6.2 MB, behind the classifier bucket on the process group's stream
        if not g_def:
            self.g_opt.step()
        ops.debug_stamp(8)
        with torch.cuda.stream(side):           # (issued after the generator's backward: the classifier's Adam sits in front of it)
            # (data parallel: nothing is deferred - the feature bucket goes out right behind this call and must be complete)
            deferred = [] if (cfg.KERNEL.DEFER_D_WGRAD and not self.dp) else None
            self._d_features(deferred)
            ar_f = sdist.AsyncAllReduce(self._d_buckets[1], self.pg, force=True) if self.one_graph_dp else None
            ops.debug_stamp(5)
            for ev, launch, tensors in (g_def or ()):
                side.wait_event(ev)
                if not torch.cuda.is_current_stream_capturing():     # eager: main's allocator must not re-use these blocks before `side` is done
                    for t in tensors:
                        if t is not None:
                            t.record_stream(side)
                launch()
        # The side branch is the longer one (D(gt) forward + two backward passes against one generator backward).  The conv weight
        # gradients of its last pass are leaves of that chain: they run HERE, on the generator's stream, which would otherwise idle
        # until the join - each behind the event of its dy.  Same kernels, same arguments, same accumulation order per parameter
        # (the D(sr) pass wrote its gradients before the D(gt) pass produced its first dy).
        capturing = torch.cuda.is_current_stream_capturing()
        for ev, launch, tensors in (deferred or ()):
            main.wait_event(ev)
            if not capturing:            # eager: the side stream's allocator must not hand these blocks out again before `main` is done
                for t in tensors:        # with them.  Under capture nothing is handed out again inside the graph (the closures hold the
                    if t is not None:    # tensors until every launch of the iteration is issued) - and recording the CAPTURE stream, which
                        t.record_stream(main)   # is destroyed after the capture, left the allocator with a dangling stream (segfaults in later replays)
            launch()
        main.wait_stream(self._side_d)
        if g_def:
            self.g_opt.step()                    # its last weight gradients came from the side stream
        if self.one_graph_dp:
            dp_ar["c"].wait()                    # on the origin stream
            ar_f.wait()
        if split:
            self.d_opt.step_params(0, cls0, flat_grad=self._d_flat)
            self.D.__dict__["_packs_fresh"] = False      # weights changed
        else:
            self._d_step()
        self.D.__dict__["_counters_external"] = False
        ops.debug_stamp(9)
        return vals

    def _d_two_stream_full(self):
        v = self._d_two_stream()
        self._d_step()
        return v

    def _d_step(self):
        self.d_opt.step()
        self.D.__dict__["_packs_fresh"] = False      # weights changed

    def _d_full(self):
        v = self._d_fwd_bwd()
        self._d_step()
        return v

    def close(self):
        """See WarmupEngine.close."""
        self._g_fb = self._g_op = self._d_fb = self._d_op = self._d_a = self._d_b = self._it = self._g_f = self._g_b = None
        self._g_total = None
        self._d_state = self._d_flat = self._d_buckets = None
        self.D.__dict__.pop("_last_pass", None)
        self.gt = self.lr = self.sr = None

    def _step_overlapped(self):
        """Data-parallel iteration: collectives hidden behind compute AND the discriminator half beside the generator's backward
        (the two-branch schedule of _iter_gd with the graphs cut where a collective has to go out):
            main: [G forward] -------------- [G backward] -> all-reduce(G) ................. -> [G Adam] -> join -> [D Adam]
            side:        \\-> [D fwd x2 + classifier bwd] -> all-reduce(classifier) || [D feature bwd] -> all-reduce(features)"""
        main = torch.cuda.current_stream()
        did_d = self.batch_num % self.config.SOLVER.D_UPDATE_INTERVAL == 0
        self._g_f()
        if did_d:
            if self._side_d is None:
                self._side_d = torch.cuda.Stream()
            side = self._side_d
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._d_a()
                ar_c = sdist.AsyncAllReduce(self._d_buckets[0], self.pg, force=True)          # classifier: 75.5 MB, under the feature backward
                self._d_b()
                ar_f = sdist.AsyncAllReduce(self._d_buckets[1], self.pg, force=True)
        self._g_b()
        ar_g = sdist.AsyncAllReduce(sdist.module_flat_grad(self.G), self.pg_g, force=True)
        if ar_g.flat is None:                                    # gradients not in one flat buffer: the generic path
            sdist.allreduce_module_grads(self.G, self.pg_g, force=True)
        ar_g.wait()
        self._g_op()
        if did_d:
            main.wait_stream(side)
            ar_c.wait()
            ar_f.wait()
            self._d_op()
        self.batch_num += 1
        return self.loss_values, (self.d_loss if did_d else None)

    def step(self, gt, lr):
        _load_inputs(self, gt, lr)
        if self.overlap:
            return self._step_overlapped()
        if self._it is not None:
            did_d = self.batch_num % self.config.SOLVER.D_UPDATE_INTERVAL == 0
            (self._it if did_d else self._g_fb)()
            self.batch_num += 1
            return self.loss_values, (self.d_loss if did_d else None)
        self._g_fb()
        if self.dp:
            sdist.allreduce_module_grads(self.G, self.pg, force=True)
            self._g_op()
        did_d = False
        if self.batch_num % self.config.SOLVER.D_UPDATE_INTERVAL == 0:
            self._d_fb()
            if self.dp:
                sdist.allreduce_module_grads(self.D, self.pg, buckets=2 if self.config.DIST.BUCKET_D else 1, force=True)
                self._d_op()
            did_d = True
        self.batch_num += 1
        return self.loss_values, (self.d_loss if did_d else None)

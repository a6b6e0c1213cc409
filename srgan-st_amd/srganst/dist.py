"""Data parallelism: one process per GPU, gradient averaging with RCCL over xGMI.

The reference has no distributed code at all; this is the subsystem the
build adds.  Semantics (standard DDP): every rank runs the same step on its own [B] batch,
BatchNorm statistics stay rank-local, parameter gradients are averaged over ranks before the
optimizer step, so parameters stay bit-identical across ranks.

Payload: G = 1,547,350 fp32 (6.2 MB) after every G backward; D = 23,563,649 fp32 (94 MB) after
every D backward.  xGMI is point-to-point, so collectives are per-link bound: few, large,
flat buffers - G goes out as ONE all-reduce; D as two buckets (classifier 18.9 M / features 4.7 M
floats).  engine.TrainEngine overlaps them with compute: the classifier bucket travels under the feature stack's backward, the
generator's message goes out when its backward ends - on the default communicator BEHIND the discriminator's buckets (collectives of
one process group run in issue order), or on a communicator of its own with DIST.G_OWN_GROUP (off: untested here) - (AsyncAllReduce; the collectives run on the
process groups' own streams).  All of it is UNMEASURED at N > 1 here (one GPU per call).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*) -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def is_rccl(pg=None) -> bool:
    """True when gradients travel over RCCL (backend "nccl"): its collectives can be captured into a hipGraph."""
    return td.is_available() and td.is_initialized() and td.get_backend(pg) == "nccl"


def world_size(pg=None) -> int:
    return td.get_world_size(pg) if td.is_available() and td.is_initialized() else 1


def bucket_slices(numels, buckets: int):
    """Split a parameter list into `buckets` contiguous groups of roughly equal payload (in list order)."""
    total = sum(numels)
    out, start, acc, target = [], 0, 0, total / max(buckets, 1)
    for i, n in enumerate(numels):
        acc += n
        if acc >= target * (len(out) + 1) and len(out) < buckets - 1:
            out.append((start, i + 1))
            start = i + 1
    out.append((start, len(numels)))
    return [s for s in out if s[0] < s[1]]


def allreduce_grads(params, pg=None, buckets: int = 1, force: bool = False) -> None:
    """p.grad <- mean over ranks of p.grad, for every parameter that has a gradient.
    force: issue the collective even with a single rank (exercises the RCCL path in tests)."""
    world = world_size(pg)
    if world == 1 and not (force and td.is_available() and td.is_initialized()):
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    handles = []
    for a, b in bucket_slices([p.grad.numel() for p in ps], buckets):
        grads = [p.grad for p in ps[a:b]]
        flat = torch._utils._flatten_dense_tensors(grads)
        h = td.all_reduce(flat, op=td.ReduceOp.SUM, group=pg, async_op=True)
        handles.append((h, flat, grads))
    for h, flat, grads in handles:
        h.wait()
        flat.mul_(1.0 / world)
        for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(f)


def allreduce_module_grads(module, pg=None, buckets: int = 1, force: bool = False) -> None:
    """Like allreduce_grads(module.parameters()), but if the module's backward left its gradients as views of one
    flat buffer (srganst graphs do: module._flat_grad) and they still live there, that buffer is all-reduced in place -
    one message (or `buckets` contiguous slices of it), no flatten / unflatten copies."""
    world = world_size(pg)
    if world == 1 and not (force and td.is_available() and td.is_initialized()):
        return
    ps = [p for p in module.parameters() if p.grad is not None]
    flat = None
    for cand in reversed(module.__dict__.get("_flat_grads", [])):
        lo, hi = cand.data_ptr(), cand.data_ptr() + cand.numel() * 4
        if ps and all(lo <= p.grad.data_ptr() < hi and p.grad.is_contiguous() for p in ps) and \
                sum((p.grad.numel() + 15) // 16 * 16 for p in ps) == cand.numel():
            flat = cand
            break
    if flat is None:
        return allreduce_grads(ps, pg, buckets, force)
    n = flat.numel()
    step = (n + buckets - 1) // buckets
    # RCCL averages inside the collective (one elementwise pass less); gloo (CPU tests) has no AVG: sum, then scale
    avg = td.get_backend(pg) == "nccl" and hasattr(td.ReduceOp, "AVG")
    handles = []
    for i in range(0, n, step):
        sl = flat[i:i + step]
        handles.append((td.all_reduce(sl, op=td.ReduceOp.AVG if avg else td.ReduceOp.SUM, group=pg, async_op=True), sl))
    for h, sl in handles:
        h.wait()
        if world > 1 and not avg:
            sl.mul_(1.0 / world)


class AsyncAllReduce:
    """One in-flight mean-all-reduce of a flat fp32 slice: issued on the process group's own stream (RCCL: a communication
    stream next to the compute stream, so kernels launched after start() overlap it), joined by wait().
    `force` issues the collective with a single rank too (exercises RCCL in tests); without a process group it is a no-op."""

    def __init__(self, flat, pg=None, force: bool = False):
        self.flat, self.pg, self.h = flat, pg, None
        self.world = world_size(pg)
        self.scale = False
        if flat is None or (self.world == 1 and not (force and td.is_available() and td.is_initialized())):
            return
        avg = td.get_backend(pg) == "nccl" and hasattr(td.ReduceOp, "AVG")
        self.h = td.all_reduce(flat, op=td.ReduceOp.AVG if avg else td.ReduceOp.SUM, group=pg, async_op=True)
        self.scale = self.world > 1 and not avg

    def wait(self):
        if self.h is not None:
            self.h.wait()
            if self.scale:
                self.flat.mul_(1.0 / self.world)
            self.h = None


def module_flat_grad(module):
    """The flat buffer the module's parameter gradients are views of (srganst graphs leave them that way), or None."""
    ps = [p for p in module.parameters() if p.grad is not None]
    for cand in reversed(module.__dict__.get("_flat_grads", [])):
        lo, hi = cand.data_ptr(), cand.data_ptr() + cand.numel() * 4
        if ps and all(lo <= p.grad.data_ptr() < hi and p.grad.is_contiguous() for p in ps) and \
                sum((p.grad.numel() + 15) // 16 * 16 for p in ps) == cand.numel():
            return cand
    return None


def broadcast_module(module, src: int = 0, pg=None) -> None:
    """Make parameters and buffers identical on every rank (rank `src` wins)."""
    if world_size(pg) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        td.broadcast(t.data, src=src, group=pg)

"""CPU, world_size 2 over gloo: the data-parallel exchange step (flat gradient all-reduce, buckets,
parameter broadcast).  Parity statement: after the exchange every rank holds the mean
of the per-rank gradients, so an optimizer step equals the single-process step on the averaged gradient."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "srgan-st_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from srganst import dist as sdist
    r, l, w = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world) and sdist.world_size() == world
    torch.manual_seed(100 + rank)                                   # ranks start different ...
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    sdist.broadcast_module(model)                                   # ... and are made identical
    w0 = [p.detach().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(6, 7, generator=g)
    model(x).square().mean().backward()
    local = [p.grad.clone() for p in model.parameters()]
    sdist.allreduce_grads(model.parameters(), buckets=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-4)
    opt.step()
    q.put((rank, [t.numpy() for t in w0], [t.numpy() for t in local], [p.grad.numpy().copy() for p in model.parameters()],
           [p.detach().numpy().copy() for p in model.parameters()]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_allreduce_grads_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0a, la, ga, pa), (_, w0b, lb, gb, pb) = res
    for a, b in zip(w0a, w0b):
        assert (a == b).all()                                        # broadcast made the replicas identical
    for a, b, x, y in zip(la, lb, ga, gb):
        assert abs((a + b) / 2 - x).max() < 1e-7 and (x == y).all()  # mean of per-rank grads, same on both ranks
    for a, b in zip(pa, pb):
        assert (a == b).all()                                        # parameters stay bit-identical after the step


def _async_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "srgan-st_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from srganst import dist as sdist
    sdist.init_from_env("gloo")
    # a module whose gradients are views of ONE flat buffer with 16-float aligned slots (what the srganst graphs leave behind)
    torch.manual_seed(5)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ps = list(model.parameters())
    offs, off = [], 0
    for p in ps:
        offs.append(off)
        off += (p.numel() + 15) // 16 * 16
    g = torch.Generator().manual_seed(10 + rank)
    flat = torch.randn(off, generator=g)
    for p, o in zip(ps, offs):
        p.grad = flat[o:o + p.numel()].view(p.shape)
    model.__dict__["_flat_grads"] = [torch.zeros(3), flat]
    found = sdist.module_flat_grad(model)
    assert found is flat
    local = flat.clone()
    cut = offs[2]                                                   # two buckets, the later parameters first (engine order)
    h1 = sdist.AsyncAllReduce(flat[cut:], None)
    other = torch.full((4,), float(rank))                           # work issued between start and wait
    other = other * 2 + 1
    h0 = sdist.AsyncAllReduce(flat[:cut], None)
    h1.wait()
    h0.wait()
    h0.wait()                                                       # idempotent
    noop = sdist.AsyncAllReduce(None, None)
    noop.wait()
    q.put((rank, local.numpy(), flat.numpy().copy(), [p.grad.numpy().copy() for p in ps], other.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_async_bucket_allreduce_world2():
    """The overlapped exchange of engine.TrainEngine (AsyncAllReduce on slices of the module's flat gradient buffer): every rank
    ends with the mean of the per-rank buffers, in place (the parameter gradients are views of it)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, la, fa, ga, oa), (_, lb, fb, gb, ob) = res
    assert (fa == fb).all() and abs((la + lb) / 2 - fa).max() < 1e-7
    off = 0
    for x in ga:                                                     # gradients still alias the (now averaged) flat buffer
        assert (x.ravel() == fa[off:off + x.size]).all()
        off += (x.size + 15) // 16 * 16
    assert (oa == 1).all() and (ob == 3).all()

"""FlatAdam (srganst/optim.py + csrc/optim.hip) against torch.optim.Adam: same trajectory, same state_dict layout,
schedulers and the stock-path fallback.  Tolerance: a few fp32 ulps per step (bias corrections are computed in double
in both; only FMA contraction differs)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.a = torch.nn.Parameter(torch.randn(64, 3, 9, 9, generator=g))
        self.slope = torch.nn.Parameter(torch.tensor([0.25]))
        self.b = torch.nn.Parameter(torch.randn(3, generator=g))
        self.c = torch.nn.Parameter(torch.randn(37, 5, generator=g))


def _grads(net, step):
    g = torch.Generator().manual_seed(100 + step)
    return [torch.randn(p.shape, generator=g).cuda() * (10.0 ** (step % 3 - 1)) for p in net.parameters()]


@pytest.mark.parametrize("capturable", [False, True])
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_matches_torch_adam(capturable, wd):
    from srganst import ops
    from srganst.optim import FlatAdam
    ref_net, net = _Net().cuda(), _Net().cuda()
    ref = torch.optim.Adam(ref_net.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-4, weight_decay=wd)
    opt = FlatAdam(net, lr=1e-2, betas=(0.9, 0.999), eps=1e-4, weight_decay=wd, capturable=capturable)
    sched_ref = torch.optim.lr_scheduler.MultiStepLR(ref, milestones=[3], gamma=0.5)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3], gamma=0.5)
    names = [n for n, _ in net.named_parameters()]
    for step in range(6):
        gs = _grads(net, step)
        for p, g in zip(ref_net.parameters(), gs):
            p.grad = g.clone()
        views = ops.flat_grads(net, names, list(net.parameters()))     # what the HIP backward passes hand to autograd
        for (n, p), g in zip(net.named_parameters(), gs):
            views[n].copy_(g)
            p.grad = views[n]
        assert opt._flat_grad() is not None
        ref.step()
        opt.step()
        sched_ref.step()
        sched.step()
        for (n, p), q in zip(net.named_parameters(), ref_net.parameters()):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (step, n)
    assert float(opt.param_groups[0]["lr"]) == pytest.approx(ref.param_groups[0]["lr"])
    # state_dict: torch's layout; round trip into a fresh optimizer continues the same trajectory
    sd = copy.deepcopy(opt.state_dict())
    sd_ref = ref.state_dict()
    assert sd["state"].keys() == sd_ref["state"].keys()
    for k in sd["state"]:
        assert float(sd["state"][k]["step"]) == float(sd_ref["state"][k]["step"]) == 6
        for key in ("exp_avg", "exp_avg_sq"):          # norm-wise: single elements of exp_avg cancel to ~0
            a, b = sd["state"][k][key].double(), sd_ref["state"][k][key].double()
            assert float((a - b).norm() / b.norm()) < 1e-6
    net2 = _Net().cuda()
    net2.load_state_dict(net.state_dict())
    opt2 = FlatAdam(net2, lr=1e-2, betas=(0.9, 0.999), eps=1e-4, weight_decay=wd, capturable=capturable)
    opt2.load_state_dict(sd)
    gs = _grads(net, 7)
    for o, m in ((opt, net), (opt2, net2)):
        for p, g in zip(m.parameters(), gs):
            p.grad = g.clone()                      # plain tensors: the stock torch path on the flat state
        assert o._flat_grad() is None
        o.step()
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8)
    for p, g in zip(ref_net.parameters(), gs):
        p.grad = g.clone()
    ref.step()
    for p, q in zip(net.parameters(), ref_net.parameters()):
        assert torch.allclose(p, q, rtol=5e-6, atol=1e-7)


def test_flatten_params_keeps_module_semantics():
    from srganst import ops
    net = _Net().cuda()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat, offs, params = ops.flatten_params(net)
    assert all(o % 16 == 0 for o in offs) and flat.numel() % 16 == 0
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    net.load_state_dict({k: v * 2 for k, v in before.items()})         # in-place copy: views stay views
    assert all(p.data_ptr() == flat.data_ptr() + 4 * o for p, o in zip(params, offs))
    assert torch.equal(net.c.data, before["c"] * 2)
    assert ops.flatten_params(net)[0] is flat                          # idempotent

"""CPU: pin the oracle (our restatement) to the outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py, which
imports /root/reference in the build container.  Tolerances are those of fp32 re-association
(oneDNN conv vs batched conv, vmap vs batch) - far below the 1e-3 the north star allows.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import model as om
from oracle import st as ost
from oracle import steps as osteps


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_filter_taps(golden):
    g = golden("st_ops")
    g5, dg5 = ost.gaussian_kernel(0.5, also_dg=True)
    k17 = ost.gaussian_kernel(2.0)
    assert g5.numel() == 5 and k17.numel() == 17          # 5 / 17 taps
    assert torch.equal(g5, T(g["g5"])) and torch.equal(dg5, T(g["dg5"])) and torch.equal(k17, T(g["k17"]))


def test_structure_tensor_blocks(golden):
    g = golden("st_ops")
    S1 = ost.structure_tensor(T(g["gray1"]).unsqueeze(0), 0.5, 2.0)
    S2 = ost.structure_tensor(T(g["gray2"]).unsqueeze(0), 0.5, 2.0)
    assert rel_err(S1[0], g["S1"]) < 1e-6 and rel_err(S2[0], g["S2"]) < 1e-6
    # feed the reference's own S so the pointwise algebra is compared on identical inputs
    S1r, S2r = T(g["S1"]).unsqueeze(0), T(g["S2"]).unsqueeze(0)
    assert rel_err(ost.normalize(S1r)[0], g["N1"]) < 1e-6
    M = ost.inv_s1_x_s2(S1r, S2r, True)
    assert rel_err(M[0], g["M"]) < 1e-5
    # reference layouts: M [4,H,W]; L [H,2,W]; d [H,W]
    L = ost.eigenvalues(T(g["M"]).unsqueeze(0))
    assert rel_err(L[0], T(g["L"]).permute(1, 0, 2)) < 1e-6
    d = ost.distance(T(g["L"]).permute(1, 0, 2).unsqueeze(0))
    assert rel_err(d[0], g["d"]) < 1e-6


@pytest.mark.parametrize("case", ["noise32", "lowfreq32", "flat32", "mixed96"])
def test_st_loss_and_grad(golden, case):
    g = golden("st_loss")
    x, gt = T(g[case + "_x"]), T(g[case + "_gt"])
    loss, gx = ost.st_loss_and_grad(x, gt)
    assert abs(loss.item() - g[case + "_loss"].item()) <= 2e-5 * abs(g[case + "_loss"].item())
    # the fp32 reference itself is this far from its own fp64 run; we must be no worse than 3x
    ref_err = rel_err(g[case + "_grad"], g[case + "_grad64"])
    assert rel_err(gx, g[case + "_grad64"]) <= max(3 * ref_err, 1e-5)
    # fp64 oracle == fp64 reference
    l64, g64 = ost.st_loss_and_grad(x.double(), gt.double())
    assert abs(l64.item() - g[case + "_loss64"].item()) < 1e-6 * abs(g[case + "_loss64"].item())
    assert rel_err(g64, g[case + "_grad64"]) < 1e-5


def _state(g, prefix):
    return {k[len(prefix):]: T(g[k]).clone() for k in g.files if k.startswith(prefix) and "#" not in k}


def test_generator_small_two_steps(golden):
    g = golden("g_small_step")
    sd0 = _state(g, "state0/")
    tr = osteps.OracleTrainer(sd0, criterions=(("Pixel", 1.0), ("ST", 1.0 / 3.0)))
    gt, lr = T(g["gt"]), T(g["lr"])
    sr, losses = tr.warmup_step(gt, lr)
    assert rel_err(sr, g["sr0"]) < 1e-5
    assert abs(losses["Pixel"].item() - g["loss_pixel0"].item()) < 1e-5 * abs(g["loss_pixel0"].item())
    assert abs(losses["ST"].item() - g["loss_st0"].item()) < 1e-4 * abs(g["loss_st0"].item())
    grads = tr.g_grads()
    for k, v in grads.items():
        assert rel_err(v, g["grad0/" + k]) < 2e-3, k
    s1 = _state(g, "state1/")
    for k in s1:
        if "num_batches" in k:
            assert int(tr.g[k]) == int(s1[k])
        else:
            assert torch.allclose(tr.g[k].detach(), s1[k], rtol=1e-4, atol=2e-5), k
    tr.warmup_step(gt, lr)
    s2 = _state(g, "state2/")
    for k in s2:
        if "num_batches" not in k:
            assert torch.allclose(tr.g[k].detach(), s2[k], rtol=1e-4, atol=5e-5), k


def test_generator_full_seed0(golden):
    g = golden("g_full_seed0")
    torch.manual_seed(0)
    sd = om.init_generator_state()
    assert sum(sd[k].numel() for k in om.param_keys(sd)) == int(g["n_params"]) == 1547350  # model.py:193
    for k in [f[2:] for f in g.files if f.startswith("w/")]:
        assert torch.equal(sd[k], T(g["w/" + k])), k       # identical seeds => identical parameters
    for k in om.param_keys(sd):
        sd[k].requires_grad_(True)
    nb = {}
    sr = om.generator_forward(sd, T(g["lr"]), True, nb)
    assert rel_err(sr, g["sr"]) < 1e-5
    loss = torch.nn.functional.mse_loss(sr, T(g["gt"]))
    assert abs(loss.item() - g["loss"].item()) < 1e-5 * g["loss"].item()
    loss.backward()
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    for k in om.param_keys(sd):
        assert abs(sd[k].grad.norm().item() - norms[k]) <= 2e-3 * norms[k] + 1e-9, k
    for k in [f[2:] for f in g.files if f.startswith("g/")]:
        assert rel_err(sd[k].grad, g["g/" + k]) < 2e-3, k
    assert torch.allclose(nb["trunk.0.rcb.1.running_mean"], T(g["bn/trunk.0.rcb.1.running_mean"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(nb["trunk.0.rcb.1.running_var"], T(g["bn/trunk.0.rcb.1.running_var"]), rtol=1e-4, atol=1e-6)


def test_discriminator_full_seed0(golden):
    g = golden("d_full_seed0")
    torch.manual_seed(0)
    sd = om.init_discriminator_state()
    assert sum(sd[k].numel() for k in om.param_keys(sd)) == int(g["n_params"]) == 23563649  # model.py:194
    assert torch.equal(sd["features.0.weight"], T(g["w/features.0.weight"]))
    assert torch.equal(sd["classifier.2.weight"], T(g["w/classifier.2.weight"]))
    for k in om.param_keys(sd):
        sd[k].requires_grad_(True)
    x = T(g["x"]).clone().requires_grad_(True)
    logit = om.discriminator_forward(sd, x, True, {})
    assert torch.allclose(logit, T(g["logit"]), rtol=1e-4, atol=1e-5)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, torch.full_like(logit, 0.9))
    loss.backward()
    assert rel_err(x.grad, g["dx"]) < 2e-3
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    for k in om.param_keys(sd):
        assert abs(sd[k].grad.norm().item() - norms[k]) <= 2e-3 * norms[k] + 1e-9, k


def test_gan_iteration_small(golden):
    """One full train.py iteration (G step then D step) incl. D's BN moving during the G step."""
    g = golden("gan_small_iter")
    torch.manual_seed(0)
    d0 = om.init_discriminator_state(ch=4)               # same construction order as make_golden: D then G
    g0 = om.init_generator_state(ch=8, n_rcb=2)
    for k, v in _state(g, "d_state0/").items():          # small tensors are stored whole
        assert torch.equal(d0[k], v), k
    for k in [f for f in g.files if f.startswith("d_state0/") and f.endswith("#head4")]:
        name = k[len("d_state0/"):-len("#head4")]
        assert torch.equal(d0[name][:4], T(g[k])), name   # big ones: head + norm
    for k, v in _state(g, "g_state0/").items():
        assert torch.equal(g0[k], v), k
    tr = osteps.OracleTrainer(g0, d0, criterions=(("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0)),
                              d_update_interval=1)
    gt, lr = T(g["gt"]), T(g["lr"])
    # run the G half by hand to capture D's BN buffers in between, like the fixture does
    sr, losses, d_loss = tr.train_step(gt, lr)
    assert rel_err(sr, g["sr"]) < 1e-5
    for name in ("Adversarial", "Pixel", "ST"):
        assert abs(losses[name].item() - g["g_loss/" + name].item()) < 1e-4 * abs(g["g_loss/" + name].item()), name
    gg = tr.g_grads()
    for k, v in gg.items():
        assert rel_err(v, g["g_grad/" + k]) < 3e-3, k
    assert abs(d_loss.item() - g["d_loss"].item()) < 1e-5
    dg = tr.d_grads()
    for k, v in dg.items():
        key = "d_grad/" + k
        if key in g.files:
            assert rel_err(v, g[key]) < 3e-3, k
        else:
            assert abs(v.double().norm().item() - g[key + "#norm"].item()) < 3e-3 * g[key + "#norm"].item(), k
    for k, v in _state(g, "d_state1/").items():
        if "num_batches" in k:
            assert int(tr.d[k]) == int(v) == 3            # D(sr) in G step + D(gt) + D(sr) in D step
        else:
            assert torch.allclose(tr.d[k].detach(), v, rtol=1e-4, atol=5e-5), k
    for k, v in _state(g, "g_state1/").items():
        if "num_batches" not in k:
            assert torch.allclose(tr.g[k].detach(), v, rtol=1e-4, atol=5e-5), k

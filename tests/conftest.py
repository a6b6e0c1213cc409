"""pytest config: registers the ``gpu`` marker and puts the product package on sys.path.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol check, gloo DP tests.
``-m gpu``      : parity tests proper - HIP path (through the C-ABI) vs oracle / golden.
Nothing here reads /root/reference (it does not exist on the GPU box).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "srgan-st_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def rel_err(a, b):
    """Norm-wise relative error ||a-b|| / ||b||  (b = reference)."""
    import torch
    a = torch.as_tensor(a).detach().to(torch.float64).flatten()
    b = torch.as_tensor(b).detach().to(torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---- fp64-truth criterion for gradients (north star: 1e-3 relative fp32) --------------------------------------------
# A gradient is accepted when its error against the SAME arithmetic evaluated in fp64 (the oracle run in double) is within
# 1e-3, or - for quantities whose fp32 evaluation is itself farther than that from the truth (cancelling sums such as the
# scalar PReLU slopes) - within 3x the error the reference arithmetic makes in fp32 (oracle fp32 vs oracle fp64).  No
# hand-set per-parameter numbers; same rule as tests/test_st_loss_gpu.py.
TRUTH_FLOOR = 1e-3
TRUTH_FACTOR = 3.0


def truth_bound(ref32, ref64):
    return max(TRUTH_FLOOR, TRUTH_FACTOR * rel_err(ref32, ref64))


def assert_fp64_truth(name, hip, ref32, ref64, report=None):
    e_hip, bound = rel_err(hip, ref64), truth_bound(ref32, ref64)
    if report is not None:
        report.append((name, e_hip, rel_err(ref32, ref64)))
    assert e_hip <= bound, f"{name}: |hip - fp64| = {e_hip:.3e} > max(1e-3, 3 x |fp32 oracle - fp64| = {bound:.3e})"


def oracle_grads(forward_loss, sd, dtype, inputs=()):
    """Runs `forward_loss(sd_leaf, *inputs_in_dtype) -> (loss, outputs...)` of the CPU oracle in `dtype` and returns
    (loss, {param: grad}, [input grads], outputs)."""
    import torch
    from oracle import model as om
    sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    keys = om.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    ins = [t.detach().clone().to(dtype).requires_grad_(rg) for t, rg in inputs]
    out = forward_loss(sd, *ins)
    loss, rest = (out[0], out[1:]) if isinstance(out, tuple) else (out, ())
    loss.backward()
    return loss.detach(), {k: sd[k].grad for k in keys}, [t.grad for t in ins], [r.detach() for r in rest]


@pytest.fixture(autouse=True)
def _sst_env_switches(monkeypatch):
    """libsrganst.so reads each SST_* dev switch from the environment ONCE (csrc/api.hip: sst_env); tests toggle them with
    monkeypatch.setenv / delenv, so the library's table is dropped at the start of every test (the previous test's values were
    restored after its body) and after every change made through monkeypatch."""
    def reload():
        mod = sys.modules.get("srganst._abi")
        if mod is not None:
            mod.reload_env()
    reload()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_reload(name, value, prepend=None):
        setenv(name, value, prepend)
        reload()

    def delenv_reload(name, raising=True):
        delenv(name, raising)
        reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv_reload, delenv_reload
    yield
    reload()

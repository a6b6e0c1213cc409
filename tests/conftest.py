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

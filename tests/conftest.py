import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """Entries of d under `prefix`, prefix stripped, as torch tensors."""
    return {k[len(prefix):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b, floor=1e-5):
    """max|a-b| / max(max|b|, floor) -- the tolerance metric of DESIGN.md (<= 1e-4 in
    fp32).  `floor` keeps mathematically-zero tensors (e.g. the gradient of a BN shift
    that feeds another train-mode BN) from dividing round-off by round-off."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    denom = max(float(b.abs().max()), floor)
    return float((a - b).abs().max()) / denom


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2 -- used for the bf16-operand path, where a ReLU6 mask can flip on
    elements that sit on the 0 / 6 boundary (an O(1) change of single gradient entries)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def assert_grads_close(got, ref, tol, what=""):
    """Per-tensor rel_err with the floor tied to the model's overall gradient scale
    (1e-3 of the largest |grad|), so exact-zero gradients compare at round-off."""
    scale = max(float(np.abs(np.asarray(v)).max()) for v in ref.values())
    floor = 1e-3 * scale
    for k, r in ref.items():
        e = rel_err(got[k], r, floor=floor)
        assert e < tol, f"{what}{k}: rel_err {e:.3e} >= {tol}"

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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        # a missing fixture is a FAILURE, not a skip: every reference-pinned test would otherwise vanish with exit code 0
        pytest.fail(f"golden fixture tests/golden/{name}.npz is missing (regenerate with tests/golden/make_golden.py where "
                    "/root/reference exists; the fixtures are committed and travel with the repo)", pytrace=False)
    z = np.load(path, allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" and z[k].ndim > 0 else z[k]) for k in z.files}


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden

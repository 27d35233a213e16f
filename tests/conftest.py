import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>.npz (plain arrays only, allow_pickle stays False)."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture
def golden():
    return load_golden


def rel_l2(a, b):
    """Norm-relative error ||a-b|| / ||b|| (element-wise rel-err is meaningless
    for entries that are ~0 such as d/dc of the flat stop surface)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / nb) if nb > 0 else float(np.linalg.norm(a))

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "nr_golden.npz"))


def golden_case(golden, name):
    pre = f"kern/{name}/"
    return {k[len(pre):]: golden[k] for k in golden.files if k.startswith(pre)}


def golden_case_names(golden):
    return sorted({k.split("/")[1] for k in golden.files if k.startswith("kern/")})

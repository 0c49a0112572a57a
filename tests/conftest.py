import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


class _Golden(dict):
    """Both fixture files as one read-only mapping: nr_golden.npz (known answers of the reference's own tests and
    vectors made by its imported pure-torch modules, tests/golden/make_golden.py) and kern_golden.npz (kern/*: the
    reference's rasterizer kernels run on the device, tests/golden/make_golden_kern.py)."""
    @property
    def files(self):
        return list(self.keys())


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    g = _Golden()
    for name in ("nr_golden.npz", "kern_golden.npz"):
        with np.load(os.path.join(ROOT, "tests", "golden", name)) as z:
            g.update({k: z[k] for k in z.files})
    return g


def golden_case(golden, name):
    pre = f"kern/{name}/"
    return {k[len(pre):]: golden[k] for k in golden.files if k.startswith(pre)}


def golden_case_names(golden):
    return sorted({k.split("/")[1] for k in golden.files if k.startswith("kern/")})

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


@pytest.fixture(params=["binned", "bidding"])
def coverage(request):
    """Runs the test once per form of the forward's coverage (d3m_set_coverage_form): per-tile face lists
    (k_bin_* -> k_raster_tiles, the form of big batches: bench.py's 32 views) and bidding into a 64-bit z-buffer
    (k_bid_*: small batches, dense meshes).  The form is back on "auto" afterwards."""
    from deep3dmap_amd import _lib
    with _lib.coverage_form(request.param):
        yield request.param


class kernels_launched:
    """with kernels_launched() as k: ...  ->  k.names = the library kernels launched inside (d3m_timing_collect)."""

    def __enter__(self):
        from deep3dmap_amd import _lib
        _lib.collect_kernel_times()
        _lib.kernel_timing(True)
        self.names = set()
        return self

    def __exit__(self, *exc):
        from deep3dmap_amd import _lib
        self.times = _lib.collect_kernel_times()
        self.names = set(self.times)
        _lib.kernel_timing(False)
        return False


COVERAGE_KERNELS = {"binned": {"k_bin_count", "k_bin_alloc", "k_bin_fill", "k_raster_tiles"},
                    "bidding": {"k_bid_faces", "k_bid_big", "k_bid_resolve"}}


def assert_coverage_form_ran(names, form):
    """The kernels of `form` ran and none of the other form's did."""
    other = "bidding" if form == "binned" else "binned"
    assert COVERAGE_KERNELS[form] <= set(names), (form, sorted(names))
    assert not (COVERAGE_KERNELS[other] & set(names)), (form, sorted(names))

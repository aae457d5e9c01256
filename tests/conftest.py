import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The reference's host-side helper modules (mppi_numba.density / visualization / utils) are out of this
# repository's scope; the alias package forwards them to a reference checkout.  On a box without one
# (the GPU box) the notebook-flow test uses the minimal stand-ins under tests/standins/ instead.
if "MPPI_NUMBA_REFERENCE" not in os.environ:
    _ref = "/root/reference"
    os.environ["MPPI_NUMBA_REFERENCE"] = _ref if os.path.isfile(os.path.join(_ref, "mppi_numba", "density.py")) \
        else os.path.join(ROOT, "tests", "standins")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _gpu_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should report skips, not crash in hipInit
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: not a GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

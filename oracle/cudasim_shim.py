"""Make the UNMODIFIED reference (/root/reference/mppi_numba) importable on its
own CPU path (NUMBA_ENABLE_CUDASIM=1) in this container.

TEST INFRASTRUCTURE ONLY.  Runs under /opt/conda/bin/python3.9 (numba 0.54.1,
numpy 1.26.4); the system python has no numba.  Nothing in the product
(mppi_numba_amd/) imports this file; it exists so that tests/golden/*.npz can be
regenerated from the reference's own code (see oracle/gen_golden.py).

Import this module BEFORE anything from the reference.  What it works around
(none of it changes reference arithmetic):
  * numba 0.54 refuses numpy > 1.20 and its `_internal` C extension is
    ABI-incompatible with numpy 1.26 -> spoof the version during import and
    stub the extension (the CUDA simulator never touches ufunc internals);
  * the 0.54 simulator has no `cuda.get_current_device`, which
    mppi_numba/config.py:9-12 calls at import time;
  * `cuda.jit(max_registers=...)` (mppi_numba/mppi.py:761) is not accepted by
    the simulator;
  * `np.float` (mppi_numba/mppi.py:32-33) was removed from numpy.
"""
import os
import sys
import types
import warnings

os.environ["NUMBA_ENABLE_CUDASIM"] = "1"
os.environ["NUMBA_DISABLE_JIT"] = "1"

import numpy as np  # noqa: E402

REFERENCE_ROOT = os.environ.get("MPPI_REFERENCE_ROOT", "/root/reference")

_true_version = np.__version__
np.__version__ = "1.20.3"


class _Opaque:
    def __init__(self, *args, **kwargs):
        pass


_fake_internal = types.ModuleType("numba.np.ufunc._internal")
_fake_internal._DUFunc = _Opaque
_fake_internal.PyUFunc_None = -1
_fake_internal.PyUFunc_Zero = 0
_fake_internal.PyUFunc_One = 1
_fake_internal.PyUFunc_ReorderableNone = -2
_fake_internal.fromfunc = lambda *a, **k: None
sys.modules["numba.np.ufunc._internal"] = _fake_internal

import numba  # noqa: E402,F401
from numba import cuda  # noqa: E402

np.__version__ = _true_version


class _SimulatedDevice:
    # Limits of the GPUs the reference was written for; only used for launch
    # geometry decisions (config.py:10-12), never for arithmetic.
    MAX_THREADS_PER_BLOCK = 1024
    MAX_BLOCK_DIM_X = 1024
    MAX_GRID_DIM_X = 2 ** 31 - 1


if not hasattr(cuda, "get_current_device"):
    cuda.get_current_device = lambda: _SimulatedDevice()

if not hasattr(np, "float"):
    np.float = float

_orig_jit = cuda.jit


def _jit_without_register_cap(*args, **kwargs):
    kwargs.pop("max_registers", None)
    return _orig_jit(*args, **kwargs)


cuda.jit = _jit_without_register_cap

# xoroshiro128+ is uint64 arithmetic that wraps by design
warnings.filterwarnings("ignore", category=RuntimeWarning)

if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)


def load_barebone_namespace():
    """Exec the class/kernels cells of barebone_mppi_numba.ipynb (cells 1-3)
    and return the resulting namespace (has Config, MPPI_Numba)."""
    import json

    with open(os.path.join(REFERENCE_ROOT, "barebone_mppi_numba.ipynb")) as fh:
        nb = json.load(fh)
    ns = {"__name__": "barebone_reference"}
    code_cells = [c for c in nb["cells"] if c["cell_type"] == "code"]
    for cell in code_cells[:3]:
        src = "".join(cell["source"])
        keep = [ln for ln in src.splitlines()
                if not ln.lstrip().startswith("%") and "matplotlib" not in ln
                and "plt." not in ln]
        exec(compile("\n".join(keep), "<barebone cell>", "exec"), ns)
    return ns

"""Alias package: lets code written against mit-acl/mppi_numba import the MI355X
implementation unchanged.

    from mppi_numba.config import Config            -> mppi_numba_amd.config
    from mppi_numba.mppi import MPPI_Numba          -> mppi_numba_amd.mppi
    from mppi_numba.terrain import TDM_Numba, ...   -> mppi_numba_amd.terrain

The three modules above are the hot path this repository replaces.  The reference's
notebooks also import its host-side helpers (`mppi_numba.density`,
`mppi_numba.visualization`, `mppi_numba.utils`: sample generators and matplotlib
drawing, no device code).  They resolve, in this order, to

  1. the reference's own files, when a checkout is reachable -- the directory named by
     the environment variable MPPI_NUMBA_REFERENCE (checkout root or its `mppi_numba`
     directory), or any other `mppi_numba` package directory further down `sys.path`
     (e.g. PYTHONPATH=/root/repo:/root/reference).  Nothing is copied: the directory
     is appended to this package's `__path__`, so `import mppi_numba.density` loads the
     reference's density.py while `mppi_numba.mppi` stays the HIP engine;
  2. otherwise they do not resolve: this repository replaces the hot path, not the reference's
     test-data generators and plotting.  The ImportError names
     MPPI_NUMBA_REFERENCE.  (The test suite points MPPI_NUMBA_REFERENCE at minimal stand-ins under
     tests/standins/ on boxes without the reference, so that the notebooks' flow still runs there.)
"""
import importlib
import os
import sys

from mppi_numba_amd import config, mppi, terrain  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_ENGINE = ("config", "mppi", "terrain")      # replaced by the HIP engine, never forwarded
_HOST_HELPERS = ("density", "visualization", "utils")

for _name in _ENGINE:
    sys.modules[__name__ + "." + _name] = getattr(sys.modules["mppi_numba_amd"], _name)


def _reference_package_dir():
    """Directory of the reference's `mppi_numba` package, or None."""
    def is_reference_pkg(d):
        return (os.path.isfile(os.path.join(d, "density.py")) and
                os.path.isfile(os.path.join(d, "visualization.py")) and
                os.path.realpath(d) != os.path.realpath(_HERE))

    hint = os.environ.get("MPPI_NUMBA_REFERENCE")
    candidates = []
    if hint:
        candidates += [hint, os.path.join(hint, "mppi_numba")]
    for entry in sys.path:
        candidates.append(os.path.join(entry or os.getcwd(), "mppi_numba"))
    for d in candidates:
        if os.path.isdir(d) and is_reference_pkg(d):
            return d
    return None


reference_dir = _reference_package_dir()
if reference_dir is not None:
    # the import system searches __path__ in order: this directory holds no density.py, the
    # reference's does.  config/mppi/terrain are already in sys.modules and are never searched.
    __path__.append(reference_dir)
else:
    import importlib.abc

    class _NoReferenceFinder(importlib.abc.MetaPathFinder):
        def find_spec(self, fullname, path=None, target=None):
            head, _, tail = fullname.rpartition(".")
            if head == __name__ and tail in _HOST_HELPERS:
                raise ImportError(
                    "%s is one of mit-acl/mppi_numba's host-side helper modules (sample generators, plotting), which "
                    "this package does not replace: point MPPI_NUMBA_REFERENCE at a checkout of the reference (or put "
                    "it behind this repository on PYTHONPATH)" % fullname)
            return None

    sys.meta_path.append(_NoReferenceFinder())

host_helpers_from = reference_dir  # None: the helper modules are not importable

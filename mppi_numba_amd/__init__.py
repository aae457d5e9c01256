"""mppi_numba_amd -- MI355X (gfx950) MPPI rollout engine behind the class surface
of mit-acl/mppi_numba.

    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba, TractionGrid, Terrain

(`import mppi_numba` resolves to the same modules through the thin alias package
at the repo root, so the reference's notebooks run unchanged.)

Python here is host glue only: every device operation goes through the C ABI of
libmppi_hip.so (include/mppi_hip.h).  There is no CPU fallback.
"""
__all__ = ["config", "mppi", "terrain", "barebone"]

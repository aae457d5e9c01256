"""Alias package: lets code written against mit-acl/mppi_numba
(`from mppi_numba.config import Config`, `from mppi_numba.mppi import MPPI_Numba`,
`from mppi_numba.terrain import TDM_Numba, TractionGrid`) import the MI355X
implementation unchanged."""
import sys

from mppi_numba_amd import config, mppi, terrain  # noqa: F401

sys.modules[__name__ + ".config"] = config
sys.modules[__name__ + ".mppi"] = mppi
sys.modules[__name__ + ".terrain"] = terrain

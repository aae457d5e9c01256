"""Host-side stand-in for the reference's `mppi_numba/utils.py` (angle wrapping helpers,
reference utils.py:5-31).  See mppi_numba/__init__.py for when this one is used."""
import numpy as np

_TWO_PI = 2.0 * np.pi


def normalize_angle(th):
    """Wrap a scalar angle into (-pi, pi]."""
    th = (th % _TWO_PI + _TWO_PI) % _TWO_PI
    return th - _TWO_PI if th > np.pi else th


def normalize_angle_np(th):
    """Wrap an array of angles into (-pi, pi] (returns a new array)."""
    th = (np.asarray(th, dtype=float) % _TWO_PI + _TWO_PI) % _TWO_PI
    return np.where(th > np.pi, th - _TWO_PI, th)

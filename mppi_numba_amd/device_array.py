"""Minimal stand-in for numba's DeviceNDArray as the reference's users see it:
`.shape`, `.dtype`, `.copy_to_host()`.  Device memory itself is owned by the
C library; this object only knows how to fetch it."""
import numpy as np


class DeviceArray(object):
    def __init__(self, shape, dtype, fetch):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._fetch = fetch

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def __len__(self):
        return self.shape[0]

    def copy_to_host(self):
        out = np.ascontiguousarray(self._fetch(), dtype=self.dtype)
        return out.reshape(self.shape)

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s)" % (self.shape, self.dtype)


class HostMirror(DeviceArray):
    """Device array whose content is only ever written by the host (maps, bin
    values): the host copy is authoritative, so fetching needs no transfer."""

    def __init__(self, host_array):
        arr = np.array(host_array, copy=True)
        super().__init__(arr.shape, arr.dtype, lambda: arr.copy())

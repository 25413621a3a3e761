"""Host-pinned and device memory for callers that hold no framework of their own (the pipeline itself needs none): numpy views over
hipHostMalloc memory - frames handed to Pipeline.step from them are copied by DMA - and plain device buffers (vbt_host_alloc /
vbt_device_alloc, include/vbt_hip.h)."""
import ctypes
import weakref

import numpy as np

from . import _lib


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (hipHostMalloc); freed when the array and every view of it are gone."""
    dtype = np.dtype(dtype)
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    p = ctypes.c_void_p()
    _lib.check(_lib.lib().vbt_host_alloc(nbytes, ctypes.byref(p)))
    buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
    weakref.finalize(buf, _lib.lib().vbt_host_free, ctypes.c_void_p(p.value))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class DeviceBuffer:
    """`nbytes` of device memory; `.ptr` is the raw device pointer Pipeline.step / step_runs accept."""

    def __init__(self, nbytes, device=0):
        self.nbytes, self.device = int(nbytes), int(device)
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().vbt_device_alloc(self.device, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self._fin = weakref.finalize(self, _lib.lib().vbt_device_free, ctypes.c_void_p(self.ptr))

    @classmethod
    def from_host(cls, array, device=0):
        a = np.ascontiguousarray(array)
        self = cls(a.nbytes, device)
        _lib.check(_lib.lib().vbt_memcpy(self.ptr, a.ctypes.data, a.nbytes, 0))
        return self

    def to_host(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _lib.check(_lib.lib().vbt_memcpy(out.ctypes.data, self.ptr, out.nbytes, 1))
        return out

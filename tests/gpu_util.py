"""Small helpers for the -m gpu tests: device buffers through the C ABI's own panda_malloc/panda_memcpy."""
import ctypes as C

import numpy as np

from panda_amd import gpu_ffi as ffi


def has_gpu() -> bool:
    try:
        n = C.c_int(0)
        return ffi.load().panda_get_device_number(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


class DeviceBuffer:
    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        ffi.check(ffi.load().panda_malloc(C.byref(self.ptr), max(self.nbytes, 16)), "CreateContextError")

    @classmethod
    def from_host(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        ffi.check(ffi.load().panda_memcpy(buf.ptr, C.c_void_p(a.ctypes.data), a.nbytes), "CreateContextError")
        return buf

    def to_host(self, dtype=np.uint32, nbytes=None, offset=0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        ffi.check(ffi.load().panda_memcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr.value + offset), nbytes), "CreateContextError")
        return out

    def free(self):
        if self.ptr:
            ffi.load().panda_free(self.ptr)
            self.ptr = C.c_void_p()


NULL_STREAM = ffi.PandaStream()

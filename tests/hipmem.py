"""Device buffers for tests through the HIP runtime the product library itself is linked against
(ctypes on its soname: the dynamic loader hands back the copy libaspect_amd.so already loaded).
torch is deliberately not used here: its wheel bundles a HIP runtime of its own, and whichever of the
two is loaded second sees no GPU."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7", mode=C.RTLD_GLOBAL)
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


def free_bytes():
    """Free device memory of the current device (hipMemGetInfo), after draining the device."""
    free, total = C.c_size_t(), C.c_size_t()
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
    _ok(hip().hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value


class DeviceArray:
    """n doubles on the current device."""

    def __init__(self, n):
        self.n = int(n)
        self.ptr = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(self.ptr), max(self.n, 1) * 8), "hipMalloc")

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        _ok(hip().hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), self.n * 8, 1), "hipMemcpy H2D")
        return self

    def fill_bytes(self, byte):
        _ok(hip().hipMemset(self.ptr, byte, self.n * 8), "hipMemset")
        return self

    def download(self):
        out = np.empty(self.n)
        _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        _ok(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 8, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

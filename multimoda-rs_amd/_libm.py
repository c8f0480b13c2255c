"""glibc ``sincos`` for the host-side Python code.

Where the reference evaluates ``angle.cos()`` and ``angle.sin()`` of the same value in one function,
LLVM on x86_64-linux-gnu emits one ``sincos`` libcall, and glibc's sincos is not bit-identical to
separate sin()/cos() for every argument (1 ulp apart for some |x| > 2.4).  Everything in this
package that mirrors such a pair goes through this helper (the C++ host and the C oracle call
``sincos`` directly).
"""
import ctypes
import ctypes.util

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None


def sincos(x: float):
    """(sin(x), cos(x)) from one glibc sincos call."""
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(float(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value

"""ctypes binding of include/r3g.h.  There is no Python/CPU fallback: a missing or unloadable
libr3g.so, or a missing GPU, raises immediately."""
import ctypes
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libr3g.so")

R3G_ERR_LEVEL_RANGE = -10
R3G_ERR_NO_SURFACE = -11


class R3GError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libr3g error %d: %s" % (code, msg))
        self.code = code


class LevelRangeError(ValueError):
    """skimage: ValueError('Surface level must be within volume data range.')"""


class NoSurfaceError(RuntimeError):
    """skimage: RuntimeError('No surface found at the given iso value.')"""


# every symbol include/r3g.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_I64P = ctypes.POINTER(ctypes.c_int64)
SYMBOLS = {
    "r3g_version": (_I, []),
    "r3g_last_error": (ctypes.c_char_p, []),
    "r3g_create": (_I, [_I, ctypes.POINTER(_P)]),
    "r3g_destroy": (None, [_P]),
    "r3g_mc_count": (_I, [_P, _P, _I, _I, _I, _D, _I, _I64P, _I64P, _P]),
    "r3g_mc_emit": (_I, [_P, _P, _P, _P, _I, _P]),
}

_LIB = None
_CTX = {}


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libr3g.so not built: run `python 3d-re-gen_amd/build.py` "
                              "(or __graft_entry__.build()); there is no fallback path")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc == 0:
        return
    msg = lib().r3g_last_error().decode("utf-8", "replace")
    if rc == R3G_ERR_LEVEL_RANGE:
        raise LevelRangeError(msg)
    if rc == R3G_ERR_NO_SURFACE:
        raise NoSurfaceError(msg)
    raise R3GError(rc, msg)


def context(device=0):
    """The process-wide r3g_ctx of a device (created on first use)."""
    if device not in _CTX:
        h = _P()
        check(lib().r3g_create(int(device), ctypes.byref(h)))
        _CTX[device] = h
    return _CTX[device]

"""ctypes loader for tests/emu/mc_emu.cpp (host emulation of the HIP launch structure; test-only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libr3g_emu.so")
        src = os.path.join(_HERE, "mc_emu.cpp")
        hdr = os.path.join(_ROOT, "3d-re-gen_amd", "csrc", "mc_cell.h")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w",
                                   "-I" + os.path.join(_ROOT, "3d-re-gen_amd", "csrc"), "-o", so, src])
        lib = ctypes.CDLL(so)
        lib.r3g_emu_mc.restype = ctypes.c_int
        lib.r3g_emu_mc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_uint)]
        lib.r3g_emu_free.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def marching_cubes(vol, level, classic=False, xform=None, reversed_faces=True):
    """-> (verts, faces, flags).  xform = (grid_size[3], bbox_size[3], bbox_min[3]) or None."""
    vol = np.ascontiguousarray(vol, np.float32)
    pv, pf = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf, fl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint()
    xf = None
    if xform is not None:
        xf = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64) for a in xform]))
    rc = _lib().r3g_emu_mc(vol.ctypes.data, *vol.shape, float(level), int(classic),
                           xf.ctypes.data if xf is not None else None, int(reversed_faces),
                           ctypes.byref(pv), ctypes.byref(pf), ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(fl))
    assert rc == 0, rc
    v = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (max(nv.value, 1), 3))[:nv.value].copy()
    f = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)), (max(nf.value, 1), 3))[:nf.value].copy()
    _lib().r3g_emu_free(pv)
    _lib().r3g_emu_free(pf)
    return v, f, fl.value

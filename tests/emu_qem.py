"""ctypes loader for tests/emu/qem_emu.cpp (host run of the product's edge-collapse bodies; test-only)."""
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
        so = os.path.join(_HERE, "libr3g_qem_emu.so")
        src = os.path.join(_HERE, "qem_emu.cpp")
        csrc = os.path.join(_ROOT, "3d-re-gen_amd", "csrc")
        deps = [src, os.path.join(csrc, "qem_core.h"), os.path.join(csrc, "qem_driver.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w",
                                   "-I" + csrc, "-o", so, src])
        lib = ctypes.CDLL(so)
        lib.r3g_emu_qem.restype = ctypes.c_int
        lib.r3g_emu_qem.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
        _LIB = lib
    return _LIB


def reduce_faces(verts, faces, max_faces):
    """-> (verts float32 [V,3], faces int32 [F,3], rounds)"""
    v = np.ascontiguousarray(verts, np.float32).copy()
    f = np.ascontiguousarray(faces, np.int32).copy()
    nv, nf, rounds = ctypes.c_int64(len(v)), ctypes.c_int64(len(f)), ctypes.c_int(0)
    rc = _lib().r3g_emu_qem(v.ctypes.data, ctypes.byref(nv), f.ctypes.data, ctypes.byref(nf), int(max_faces),
                            ctypes.byref(rounds))
    assert rc == 0
    return v[:nv.value].copy(), f[:nf.value].copy(), rounds.value

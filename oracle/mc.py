"""ctypes front-end of oracle/mc_lewiner.c (TEST INFRASTRUCTURE ONLY).

`marching_cubes(volume, level)` mirrors what the reference path obtains from
`skimage.measure.marching_cubes(grid, level, method="lewiner")[:2]`
(upstream hy3dgen surface_extractors.MCSurfaceExtractor.run; wrapper semantics
from skimage/measure/_marching_cubes_lewiner.py:280-349), including its two
exceptions.  `hy3d_mesh(volume, level, bound, R)` adds upstream's vertex
rescale `v / (R+1) * 2*bound - bound` (float64 -> float32) and the
`faces[:, ::-1]` of export_to_trimesh.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libr3g_oracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libr3g_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.r3g_oracle_mc.restype = ctypes.c_int
        lib.r3g_oracle_mc.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        lib.r3g_oracle_free.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def marching_cubes(volume, level, use_classic=False):
    """-> (verts float32 [V,3] index space (axis0,axis1,axis2), faces int32 [F,3])."""
    if not isinstance(volume, np.ndarray) or volume.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    vol = np.ascontiguousarray(volume, np.float32)
    level = float(level)
    if level < vol.min() or level > vol.max():
        raise ValueError("Surface level must be within volume data range.")
    pv, pf = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    rc = _lib().r3g_oracle_mc(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], level,
                              int(bool(use_classic)), ctypes.byref(pv), ctypes.byref(pf),
                              ctypes.byref(nv), ctypes.byref(nf))
    if rc:
        raise MemoryError("oracle marching cubes failed rc=%d" % rc)
    try:
        if nv.value == 0:
            raise RuntimeError("No surface found at the given iso value.")
        verts = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)),
                                      (nv.value, 3)).copy()
        faces = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)),
                                      (nf.value, 3)).copy()
    finally:
        _lib().r3g_oracle_free(pv)
        _lib().r3g_oracle_free(pf)
    return verts, faces


def hy3d_mesh(volume, level=0.0, bound=1.01, octree_resolution=None):
    """Upstream MCSurfaceExtractor.run + export_to_trimesh on one grid [R+1]^3."""
    if octree_resolution is None:
        octree_resolution = volume.shape[0] - 1
    verts, faces = marching_cubes(volume, level)
    grid_size = np.array([int(octree_resolution) + 1] * 3)  # int64, upstream quirk: R+1 not R
    bbox_min = np.array([-bound] * 3)
    bbox_size = np.array([bound] * 3) - bbox_min
    v = (verts / grid_size * bbox_size + bbox_min).astype(np.float32)
    return v, np.ascontiguousarray(faces[:, ::-1])

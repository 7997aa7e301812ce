"""Marching cubes on the GPU: the host mirror of what hy3dgen's MCSurfaceExtractor does with
skimage (upstream surface_extractors.py; reached from reference src/2d_to_3d_models/run.py:77-84).

    verts, faces = marching_cubes(grid, level)                  # skimage's return convention
    verts, faces = extract_mesh(grid, mc_level, bounds, R)      # upstream MCSurfaceExtractor.run
                                                                # + export_to_trimesh winding
Inputs and outputs are torch CUDA tensors; the grid never leaves HBM.
"""
import ctypes

import numpy as np
import torch

from . import ffi as _l


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(grid, level, classic, xform, reverse, ctx=None):
    if not isinstance(grid, torch.Tensor) or grid.ndim != 3:
        raise ValueError("Input volume should be a 3D torch tensor.")
    if not grid.is_cuda:
        raise ValueError("r3g.mc needs a CUDA(HIP) tensor: the product has no CPU path")
    if min(grid.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    grid = grid.contiguous().float()
    dev = grid.device.index or 0
    with torch.cuda.device(dev):
        ctx = ctx if ctx is not None else _l.context(dev)
        L = _l.lib()
        nv, nf = ctypes.c_int64(), ctypes.c_int64()
        _l.check(L.r3g_mc_count(ctx, grid.data_ptr(), grid.shape[0], grid.shape[1], grid.shape[2], float(level),
                                int(bool(classic)), ctypes.byref(nv), ctypes.byref(nf), _stream_ptr()))
        verts = torch.empty((nv.value, 3), dtype=torch.float32, device=grid.device)
        faces = torch.empty((nf.value, 3), dtype=torch.int32, device=grid.device)
        xf = None
        if xform is not None:
            xf = np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).reshape(3) for a in xform]))
        _l.check(L.r3g_mc_emit(ctx, verts.data_ptr(), faces.data_ptr(), xf.ctypes.data if xf is not None else None,
                               int(bool(reverse)), _stream_ptr()))
    return verts, faces


def marching_cubes(grid, level, use_classic=False):
    """== skimage.measure.marching_cubes(grid, level, method="lewiner")[:2] (float32 [V,3] index
    space in (axis0, axis1, axis2) order, int32 [F,3]); raises ValueError / RuntimeError alike."""
    return _run(grid, level, use_classic, None, True)


def extract_mesh(grid, mc_level=0.0, bounds=1.01, octree_resolution=None, ctx=None):
    """Upstream MCSurfaceExtractor.run (+ the faces[:, ::-1] of export_to_trimesh) on one grid."""
    if octree_resolution is None:
        octree_resolution = grid.shape[0] - 1
    if isinstance(bounds, (int, float)):
        bounds = [-bounds, -bounds, -bounds, bounds, bounds, bounds]
    bmin, bmax = np.array(bounds[0:3], np.float64), np.array(bounds[3:6], np.float64)
    gs = np.array([int(octree_resolution) + 1] * 3, np.float64)  # upstream divides by R+1
    return _run(grid, mc_level, False, (gs, bmax - bmin, bmin), False, ctx)

"""Mesh cleaners on device buffers (include/r3g.h "mesh cleaners"): the GPU side of
hy3dgen.shapegen.postprocessors.  Inputs are torch CUDA tensors verts float32 [V,3] / faces int32 [F,3]; every
function returns new (verts, faces) tensors holding the compacted result (the inputs are not modified)."""
import ctypes

import torch

from . import ffi


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prepare(verts, faces):
    if not (verts.is_cuda and faces.is_cuda):
        raise ValueError("mesh buffers must live on the GPU (there is no CPU path)")
    v = verts.detach().to(torch.float32).contiguous().clone()
    f = faces.detach().to(torch.int32).contiguous().clone()
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
        raise ValueError("expected verts [V,3] and faces [F,3]")
    return v, f


def _run(fn, verts, faces, *extra):
    v, f = _prepare(verts, faces)
    nv, nf = ctypes.c_int64(v.shape[0]), ctypes.c_int64(f.shape[0])
    dev = v.device.index or 0
    # the shared context's workspace and read-back buffer: one call at a time (the entry points synchronise the stream)
    with ffi.device_lock(dev), torch.cuda.device(v.device):
        ctx = ffi.context(dev)
        ffi.check(fn(ctx, ctypes.c_void_p(v.data_ptr()), ctypes.byref(nv), ctypes.c_void_p(f.data_ptr()),
                     ctypes.byref(nf), *extra, _stream_ptr()))
    return v[:nv.value], f[:nf.value]


def remove_floaters(verts, faces, min_ratio=0.005):
    return _run(ffi.lib().r3g_mesh_remove_floaters, verts, faces, ctypes.c_double(min_ratio))


def remove_degenerate(verts, faces):
    return _run(ffi.lib().r3g_mesh_remove_degenerate, verts, faces)


def reduce_faces(verts, faces, max_faces=40000):
    """FaceReducer: quadric-error-metric edge collapse down to <= max_faces faces; closed surfaces stay closed manifolds
    of the same genus, boundaries stay boundaries (csrc/qem_core.h).  A budget the topology cannot reach (fewer faces
    than the components need) is reported with a warning, never silently."""
    v, f = _run(ffi.lib().r3g_mesh_reduce_faces, verts, faces, ctypes.c_int64(int(max_faces)))
    if f.shape[0] > max_faces:
        import warnings
        warnings.warn("r3g.meshops.reduce_faces: stopped at %d faces, above the budget of %d: no further collapse keeps "
                      "the mesh a manifold" % (f.shape[0], max_faces), RuntimeWarning)
    return v, f


def cluster_faces(verts, faces, max_faces=40000):
    """vertex clustering on a uniform grid (round 1's reducer: robust on triangle soups, does not preserve topology)"""
    return _run(ffi.lib().r3g_mesh_cluster_faces, verts, faces, ctypes.c_int64(int(max_faces)))

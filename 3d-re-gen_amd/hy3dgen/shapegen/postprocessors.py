"""Mesh cleaners with the call surface of upstream hy3dgen/shapegen/postprocessors.py
(`mesh = cleaner(mesh)`, reference src/2d_to_3d_models/run.py:93-94).  Upstream runs pymeshlab on one CPU thread;
here the mesh stays in HBM where marching cubes left it and the cleaners are HIP kernels (include/r3g.h "mesh
cleaners", csrc/mesh_kernels.hip):
  FloaterRemover        : drop connected components smaller than 0.5 % of the largest (by face count)
  DegenerateFaceRemover : drop faces with repeated vertices and unreferenced vertices
  FaceReducer           : reduce to <= max_facenum faces by quadric-error-metric edge collapse (upstream: MeshLab
                          meshing_decimation_quadric_edge_collapse with boundary / normal / topology preservation);
                          equivalence with MeshLab's sequential queue is geometric, not index-wise
There is no CPU path: a mesh that only has host arrays is uploaded first, and without a GPU the call raises.
Per-vertex colours are not carried through (marching-cubes meshes have none).
"""
from r3g import meshops
from r3g.mesh import Mesh


def _as_mesh(mesh):
    if isinstance(mesh, Mesh):
        return mesh
    import numpy as np
    return Mesh(np.asarray(mesh.vertices), np.asarray(mesh.faces))


def _apply(mesh, fn, *args):
    m = _as_mesh(mesh)
    if m.is_empty:
        return m.copy()
    v, f = fn(*m.device_buffers(), *args)
    return Mesh.from_device(v, f, m.metadata)


class FloaterRemover:
    def __call__(self, mesh, min_ratio=0.005):
        return _apply(mesh, meshops.remove_floaters, min_ratio)


class DegenerateFaceRemover:
    def __call__(self, mesh):
        return _apply(mesh, meshops.remove_degenerate)


class FaceReducer:
    def __call__(self, mesh, max_facenum=40000):
        return _apply(mesh, meshops.reduce_faces, max_facenum)

"""Mesh cleaners with the call surface of upstream hy3dgen/shapegen/postprocessors.py
(`mesh = cleaner(mesh)`, reference src/2d_to_3d_models/run.py:93-94).  Upstream implements them with
pymeshlab on the CPU; pymeshlab is not available here and these are SURVEY.md section 8(f) rank-1
"next" rows -- host-side numpy/scipy restatements of the same operations:
  FloaterRemover        : drop connected components smaller than 0.5 % of the largest (meshlab
                          'meshing_remove_connected_component_by_diameter'-like, by face count here)
  DegenerateFaceRemover : drop faces with repeated vertices and unreferenced vertices
  FaceReducer           : reduce to <= max_facenum faces (vertex clustering on a uniform grid; upstream
                          uses quadric edge collapse -- geometric, not bit-wise, equivalence)
"""
import numpy as np

from r3g.mesh import Mesh


def _as_mesh(mesh):
    if isinstance(mesh, Mesh):
        return mesh
    return Mesh(np.asarray(mesh.vertices), np.asarray(mesh.faces))


class FloaterRemover:
    def __call__(self, mesh, min_ratio=0.005):
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        m = _as_mesh(mesh).copy()
        if m.is_empty:
            return m
        f = m.faces
        n = len(m.vertices)
        rows = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
        cols = np.concatenate([f[:, 1], f[:, 2], f[:, 0]])
        _, label = connected_components(coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n)),
                                        directed=False)
        fl = label[f[:, 0]]
        counts = np.bincount(fl)
        keep = counts[fl] >= max(1, int(np.ceil(min_ratio * counts.max())))
        m.update_faces(keep)
        m.remove_unreferenced_vertices()
        return m


class DegenerateFaceRemover:
    def __call__(self, mesh):
        m = _as_mesh(mesh).copy()
        if m.is_empty:
            return m
        m.update_faces(m.nondegenerate_faces())
        m.remove_unreferenced_vertices()
        return m


class FaceReducer:
    def __call__(self, mesh, max_facenum=40000):
        m = _as_mesh(mesh).copy()
        if m.is_empty or len(m.faces) <= max_facenum:
            return m
        v, f = m.vertices, m.faces
        lo, hi = v.min(axis=0), v.max(axis=0)
        extent = max(float((hi - lo).max()), 1e-12)
        res = max(4, int(np.sqrt(max_facenum / 2.2)))      # a closed surface crossing an r^3 grid has ~2.2 r^2 faces
        for _ in range(24):
            cell = np.floor((v - lo) / extent * res).astype(np.int64).clip(0, res - 1)
            key = (cell[:, 0] * res + cell[:, 1]) * res + cell[:, 2]
            uniq, inv = np.unique(key, return_inverse=True)
            nf = inv[f]
            ok = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
            nf = nf[ok]
            # drop duplicate faces created by the clustering
            _, first = np.unique(np.sort(nf, axis=1), axis=0, return_index=True)
            nf = nf[np.sort(first)]
            if len(nf) <= max_facenum:
                break
            res = max(2, int(res * 0.9))
        cnt = np.bincount(inv, minlength=len(uniq)).astype(np.float64)
        nv = np.stack([np.bincount(inv, weights=v[:, a], minlength=len(uniq)) / cnt for a in range(3)], axis=1)
        out = Mesh(nv, nf)
        out.remove_unreferenced_vertices()
        return out

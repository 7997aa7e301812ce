"""ORACLE (test infrastructure, not product code): numpy restatement of the three mesh cleaners the reference
stage applies after marching cubes (src/2d_to_3d_models/run.py:93-94: FloaterRemover, DegenerateFaceRemover,
FaceReducer of hy3dgen.shapegen.postprocessors).

PARITY UNPINNED: upstream implements them with pymeshlab (absent here, like hy3dgen itself); the reference holds
no test or golden mesh for them.  What is restated is the operation each class name and its MeshLab filter stand
for; FaceReducer is a vertex clustering, not MeshLab's quadric edge collapse (geometric equivalence only).
The HIP kernels (3d-re-gen_amd/csrc/mesh_kernels.hip) must reproduce these functions bit for bit.

All functions take verts float32 [V,3], faces int [F,3] and return (verts float32, faces int32).
"""
import numpy as np


def _compact(v, f, keep):
    f = f[keep]
    used = np.zeros(len(v), bool)
    used[f.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return v[used], remap[f].astype(np.int32)


def face_components(faces, n_verts, by_vertex=False):
    """label per face.  Default: faces joined through shared EDGES (MeshLab's face-face adjacency, which its
    small-component selection walks): parts that touch in one vertex only are separate.  by_vertex: joined through
    shared vertices."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    F = len(f)
    if by_vertex:
        rows = np.concatenate([f[:, 0], f[:, 0]])
        cols = np.concatenate([f[:, 1], f[:, 2]])
        _, label = connected_components(coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n_verts, n_verts)),
                                        directed=False)
        return label[f[:, 0]]
    e = np.sort(np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 1).reshape(-1, 2), axis=1)     # the 3F undirected edges
    fid = np.repeat(np.arange(F), 3)
    ok = e[:, 0] != e[:, 1]                       # a collapsed side of a degenerate face joins nothing
    key, fid = e[ok, 0] * np.int64(n_verts) + e[ok, 1], fid[ok]
    order = np.argsort(key, kind="stable")
    key, fid = key[order], fid[order]
    same = key[1:] == key[:-1]                    # consecutive faces around one edge (two, or more if non-manifold)
    rows, cols = fid[:-1][same], fid[1:][same]
    _, label = connected_components(coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(F, F)), directed=False)
    return label


def remove_floaters(verts, faces, min_ratio=0.005, by_vertex=False):
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(v) == 0 or len(f) == 0:
        return v, f.astype(np.int32)
    fl = face_components(f, len(v), by_vertex)
    counts = np.bincount(fl)
    keep = counts[fl] >= max(1, int(min_ratio * counts.max()))     # truncation, as MeshLab's (unsigned)(largest * ratio)
    return _compact(v, f, keep)


def remove_degenerate(verts, faces):
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(v) == 0 or len(f) == 0:
        return v, f.astype(np.int32)
    keep = (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])
    return _compact(v, f, keep)


def reduce_faces(verts, faces, max_faces=40000):
    v32 = np.asarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(v32) == 0 or len(f) == 0 or len(f) <= max_faces:
        return v32, f.astype(np.int32)
    v = v32.astype(np.float64)
    lo, hi = v.min(axis=0), v.max(axis=0)
    extent = max(float((hi - lo).max()), 1e-12)
    res = max(4, int(np.sqrt(max_faces / 2.2)))      # a closed surface crossing an r^3 grid has ~2.2 r^2 faces
    for _ in range(24):
        cell = np.floor((v - lo) / extent * res).astype(np.int64).clip(0, res - 1)
        key = (cell[:, 0] * res + cell[:, 1]) * res + cell[:, 2]
        uniq, inv = np.unique(key, return_inverse=True)
        inv = inv.reshape(-1)
        nf = inv[f]
        ok = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
        nf = nf[ok]
        _, first = np.unique(np.sort(nf, axis=1), axis=0, return_index=True)   # first face of every vertex set
        nf = nf[np.sort(first)]
        if len(nf) <= max_faces:
            break
        res = max(2, int(res * 0.9))
    # cluster position = mean of the members, accumulated exactly in 2^-32 fixed point (order independent)
    q = np.rint(v * 4294967296.0).astype(np.int64)
    sums = np.zeros((len(uniq), 3), np.int64)
    np.add.at(sums, inv, q)
    cnt = np.bincount(inv, minlength=len(uniq)).astype(np.float64)
    pos = (sums.astype(np.float64) / 4294967296.0 / cnt[:, None]).astype(np.float32)
    return _compact(pos, nf, np.ones(len(nf), bool))

"""Geometric checks for mesh simplification tests: surface sampling, two-sided Hausdorff estimate, edge statistics."""
import numpy as np
from scipy.spatial import cKDTree


def sample_surface(v, f, per_face=4, seed=0):
    """vertices + `per_face` barycentric samples of every face (deterministic)"""
    rng = np.random.default_rng(seed)
    v = np.asarray(v, np.float64)
    f = np.asarray(f)
    b = rng.random((per_face, 3))
    b /= b.sum(1, keepdims=True)
    tri = v[f]                                              # [F,3,3]
    pts = np.einsum("sk,fkd->fsd", b, tri).reshape(-1, 3)
    return np.concatenate([v[np.unique(f)], pts, tri.mean(1)])


def point_triangle_distance(p, tri):
    """distance of points p [N,3] to triangles tri [N,3,3] (Ericson, Real-Time Collision Detection 5.1.5)"""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = p - b
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = p - c
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty_like(p)
    done = np.zeros(len(p), bool)

    def put(mask, val):
        m = mask & ~done
        out[m] = val[m]
        done[m] = True
    put((d1 <= 0) & (d2 <= 0), a)
    put((d3 >= 0) & (d4 <= d3), b)
    with np.errstate(divide="ignore", invalid="ignore"):
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0),
            b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
        den = 1.0 / (va + vb + vc)
        put(np.ones(len(p), bool), a + (vb * den)[:, None] * ab + (vc * den)[:, None] * ac)
    return np.linalg.norm(p - out, axis=1)


def one_sided(points, v, f, k=12):
    """max over `points` of the distance to the mesh (v, f): exact point-triangle distance to the k faces with the
    nearest centroids (an upper bound of the true distance that is tight for reasonably uniform meshes)"""
    v = np.asarray(v, np.float64)
    tri = v[np.asarray(f)]
    tree = cKDTree(tri.mean(1))
    k = min(k, len(tri))
    _, idx = tree.query(points, k=k)
    idx = idx.reshape(len(points), k)
    best = np.full(len(points), np.inf)
    for j in range(k):
        best = np.minimum(best, point_triangle_distance(points, tri[idx[:, j]]))
    return best


def hausdorff(v0, f0, v1, f1, per_face=3):
    """two-sided Hausdorff estimate between two meshes -> (max, mean)"""
    a = one_sided(sample_surface(v0, f0, per_face), v1, f1)
    b = one_sided(sample_surface(v1, f1, per_face), v0, f0)
    return max(a.max(), b.max()), 0.5 * (a.mean() + b.mean())


def edge_face_counts(f):
    f = np.asarray(f)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    u, cnt = np.unique(e, axis=0, return_counts=True)
    return u, cnt


def euler(v_count, f):
    u, _ = edge_face_counts(f)
    return v_count - len(u) + len(f)


def face_normals(v, f):
    v = np.asarray(v, np.float64)
    t = v[np.asarray(f)]
    return np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])


def signed_volume(v, f):
    v = np.asarray(v, np.float64)
    t = v[np.asarray(f)]
    return float(np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0)

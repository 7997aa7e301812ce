"""Shared inputs of the texture-stage tests (CPU oracle tests and GPU parity tests use the same scenes)."""
import numpy as np


def icosphere(level=2):
    """unit icosphere: vertices float32 [V, 3], faces int32 [F, 3] (closed manifold, outward winding)"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(level):
        cache, nf, verts = {}, [], list(v)

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (verts[a] + verts[b]) / 2.0
                verts.append(m / np.linalg.norm(m))
                cache[k] = len(verts) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, np.int64)
    return v.astype(np.float32), f.astype(np.int32)


def random_soup(nv, nf, seed, perspective=True):
    """random clip-space triangles, some off screen, some behind the near plane, one duplicated (depth tie)"""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.3, 1.3, (nv, 3)).astype(np.float32)
    w = (rng.uniform(0.5, 2.0, (nv, 1)) if perspective else np.ones((nv, 1))).astype(np.float32)
    pos = np.concatenate([xyz * w, w], axis=1).astype(np.float32)
    tri = np.stack([rng.permutation(nv)[:3] for _ in range(nf)]).astype(np.int32)
    tri[-1] = tri[0]                      # an exact duplicate: the smaller face index must win everywhere
    tri[-2] = [tri[1][0], tri[1][0], tri[1][2]]   # degenerate (zero area)
    return pos, tri


def ortho_clip(verts, rot, half_extent=1.2):
    """orthographic camera looking down -z of the rotated frame; +y is up -> row 0 at the top; depth in [-1, 1]"""
    p = verts.astype(np.float32) @ rot.astype(np.float32).T
    out = np.ones((len(p), 4), np.float32)
    out[:, 0] = p[:, 0] / np.float32(half_extent)
    out[:, 1] = -p[:, 1] / np.float32(half_extent)
    out[:, 2] = -p[:, 2] / np.float32(2.0 * half_extent)
    return out, p


def rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)


def face_atlas(faces, tex_size, pad=1.0):
    """per-face chart atlas: every face gets half of a square cell; -> uv float32 [3F, 2], uv_tri int32 [F, 3]"""
    nf = len(faces)
    cells = (nf + 1) // 2
    side = int(np.ceil(np.sqrt(cells)))
    cell = 1.0 / side
    m = pad / tex_size
    uv = np.zeros((3 * nf, 2), np.float32)
    for f in range(nf):
        c, upper = divmod(f, 2)
        x0, y0 = (c % side) * cell, (c // side) * cell
        if not upper:
            tri = [(x0 + m, y0 + m), (x0 + cell - 2.5 * m, y0 + m), (x0 + m, y0 + cell - 2.5 * m)]
        else:
            tri = [(x0 + cell - m, y0 + cell - m), (x0 + 2.5 * m, y0 + cell - m), (x0 + cell - m, y0 + 2.5 * m)]
        uv[3 * f:3 * f + 3] = tri
    return uv, np.arange(3 * nf, dtype=np.int32).reshape(nf, 3)

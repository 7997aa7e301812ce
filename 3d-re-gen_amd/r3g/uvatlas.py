"""Per-face chart atlas (host side, O(F)): the UV unwrap of the texture stage.

Upstream unwraps with xatlas (`hy3dgen/texgen/utils/uv_warp_utils.py: mesh_uv_wrap`), which is not available here; this is
the simplest valid unwrap instead: every face is its own chart, a right isosceles triangle in one half of a square cell of
a regular grid, with a texel margin around it.  Vertices are split per face corner (uv index = 3 * face + corner), so
there are no shared seams to keep consistent; the inpainting step fills the margins so bilinear sampling stays inside a
chart's own colours."""
import numpy as np


def face_atlas(n_faces, tex_size, pad=1.0):
    """-> uv float32 [3F, 2] in [0, 1] (v = 0 is the top row of the texture), uv_tri int32 [F, 3]"""
    nf = int(n_faces)
    cells = (nf + 1) // 2
    side = max(1, int(np.ceil(np.sqrt(cells))))
    cell = 1.0 / side
    m = float(pad) / float(tex_size)
    f = np.arange(nf)
    c, upper = f // 2, (f % 2).astype(bool)
    x0, y0 = (c % side) * cell, (c // side) * cell
    lo = np.stack([np.stack([x0 + m, y0 + m], 1), np.stack([x0 + cell - 2.5 * m, y0 + m], 1),
                   np.stack([x0 + m, y0 + cell - 2.5 * m], 1)], 1)
    hi = np.stack([np.stack([x0 + cell - m, y0 + cell - m], 1), np.stack([x0 + 2.5 * m, y0 + cell - m], 1),
                   np.stack([x0 + cell - m, y0 + 2.5 * m], 1)], 1)
    uv = np.where(upper[:, None, None], hi, lo).reshape(3 * nf, 2).astype(np.float32)
    return uv, np.arange(3 * nf, dtype=np.int32).reshape(nf, 3)


def uv_clip(uv):
    """clip-space positions that rasterise a mesh in UV space (texel of uv = round(uv * (T - 1)))"""
    out = np.zeros((len(uv), 4), np.float32)
    out[:, :2] = uv * np.float32(2.0) - np.float32(1.0)
    out[:, 3] = 1.0
    return out

"""The per-face chart atlas of the texture stage (3d-re-gen_amd/r3g/uvatlas.py; upstream unwraps with xatlas in
hy3dgen/texgen/utils/uv_warp_utils.py: mesh_uv_wrap): charts must not touch, whatever the face count and texture size."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
from r3g import uvatlas  # noqa: E402


def _cover(tri_px, T):
    """texels whose centre lies inside (or on the border of) a triangle given in texel coordinates"""
    lo = np.floor(tri_px.min(0)).astype(int)
    hi = np.ceil(tri_px.max(0)).astype(int)
    ys, xs = np.mgrid[lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
    p = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float64)
    a, b, c = tri_px.astype(np.float64)

    def edge(u, v):
        return (v[0] - u[0]) * (p[:, 1] - u[1]) - (v[1] - u[1]) * (p[:, 0] - u[0])
    e = np.stack([edge(a, b), edge(b, c), edge(c, a)], 1)
    inside = (e >= 0).all(1) | (e <= 0).all(1)
    q = p[inside].astype(int)
    q = q[(q[:, 0] >= 0) & (q[:, 0] < T) & (q[:, 1] >= 0) & (q[:, 1] < T)]
    return q


@pytest.mark.parametrize("nf,T", [(1, 64), (2, 64), (7, 128), (100, 256), (999, 512), (40000, 2048)])
def test_charts_are_disjoint_and_inside_the_texture(nf, T):
    uv, uv_tri = uvatlas.face_atlas(nf, T)
    assert uv.dtype == np.float32 and uv.shape == (3 * nf, 2) and uv_tri.dtype == np.int32 and uv_tri.shape == (nf, 3)
    assert (uv >= 0).all() and (uv <= 1).all()
    assert np.array_equal(uv_tri.reshape(-1), np.arange(3 * nf))          # one uv vertex per face corner: no shared seams
    tri = uv[uv_tri].astype(np.float64)                                   # [F, 3, 2]
    area2 = np.abs((tri[:, 1, 0] - tri[:, 0, 0]) * (tri[:, 2, 1] - tri[:, 0, 1]) -
                   (tri[:, 1, 1] - tri[:, 0, 1]) * (tri[:, 2, 0] - tri[:, 0, 0]))
    assert (area2 > 0).all()                                              # no degenerate chart
    # every chart stays inside its own grid cell, a texel margin away from the border
    cells = (nf + 1) // 2
    side = max(1, int(np.ceil(np.sqrt(cells))))
    cell = 1.0 / side
    c = np.arange(nf) // 2
    x0, y0 = (c % side) * cell, (c // side) * cell
    m = 1.0 / T
    assert (tri[:, :, 0] >= x0[:, None] + m - 1e-6).all() and (tri[:, :, 0] <= x0[:, None] + cell - m + 1e-6).all()
    assert (tri[:, :, 1] >= y0[:, None] + m - 1e-6).all() and (tri[:, :, 1] <= y0[:, None] + cell - m + 1e-6).all()
    if cell * T < 6:      # fewer than 6 texels per cell: the two charts of a cell cannot be kept apart (caller's choice of T)
        return
    # the two charts of a cell do not share a texel (checked by rasterising a sample of cells at texel centres)
    rng = np.random.default_rng(nf)
    for k in rng.choice(nf // 2, size=min(nf // 2, 40), replace=False) if nf >= 2 else []:
        a = _cover(tri[2 * k] * (T - 1), T)
        b = _cover(tri[2 * k + 1] * (T - 1), T)
        sa = {tuple(t) for t in a}
        sb = {tuple(t) for t in b}
        assert sa and sb and not (sa & sb)


def test_uv_clip_maps_the_unit_square_to_clip_space():
    uv = np.array([[0, 0], [1, 0], [0, 1], [0.5, 0.25]], np.float32)
    c = uvatlas.uv_clip(uv)
    assert c.dtype == np.float32 and c.shape == (4, 4)
    assert np.array_equal(c[:, :2], uv * 2 - 1) and (c[:, 2] == 0).all() and (c[:, 3] == 1).all()

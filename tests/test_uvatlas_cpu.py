"""The UV unwraps of the texture stage (3d-re-gen_amd/r3g/uvatlas.py; upstream unwraps with xatlas in
hy3dgen/texgen/utils/uv_warp_utils.py: mesh_uv_wrap): the per-face atlas of rounds 1-2 (charts must not touch, whatever the
face count and texture size) and the chart-based unwrap of round 3 (valid-unwrap invariants checked with the numpy rasteriser)."""
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


# ---- chart-based unwrap (round 3) -------------------------------------------------------------------------------------------
def _signed_area(uv, uv_tri):
    t = uv[uv_tri].astype(np.float64)
    return 0.5 * ((t[:, 1, 0] - t[:, 0, 0]) * (t[:, 2, 1] - t[:, 0, 1]) - (t[:, 2, 0] - t[:, 0, 0]) * (t[:, 1, 1] - t[:, 0, 1]))


def _helicoid(turns=2.5, pitch=0.05):
    th = np.linspace(0, turns * 2 * np.pi, 200)
    r = np.linspace(0.3, 1.0, 12)
    TH, R = np.meshgrid(th, r, indexing="ij")
    v = np.stack([R * np.cos(TH), R * np.sin(TH), pitch * TH], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(200 * 12).reshape(200, 12)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)


def _check_atlas(v, f, T, max_vertex_ratio):
    """invariants of a valid unwrap, checked with the numpy rasteriser of the texture oracle: inside the texture, every
    triangle keeps a positive orientation, vertices are shared inside charts, different charts never share a texel, and no
    texel is claimed twice inside a chart (texels covered by the whole atlas == the sum of the triangles' areas)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import tex_ref
    uv, uv_tri, uv_to_pos, chart = uvatlas.chart_atlas(v, f, T)
    assert uv.dtype == np.float32 and uv_tri.dtype == np.int32 and uv_tri.shape == (len(f), 3) and len(uv) == len(uv_to_pos)
    assert (uv >= 0).all() and (uv <= 1).all()
    assert np.array_equal(uv_to_pos[uv_tri], np.asarray(f))                 # a UV vertex stands for the mesh vertex of its corner
    assert len(uv) <= max_vertex_ratio * len(v)                              # seams only: no per-corner split
    area = _signed_area(uv, uv_tri)
    nondeg = np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1) > 0
    assert (area[nondeg] > 0).all()                                          # orientation kept: locally injective
    # charts are disjoint in the texture: their texel bounding boxes (grown by one texel) do not intersect
    n = int(chart.max()) + 1
    lo = np.full((n, 2), np.inf)
    hi = np.full((n, 2), -np.inf)
    cv = np.repeat(chart, 3)
    np.minimum.at(lo, cv, uv[uv_tri.reshape(-1)] * (T - 1))
    np.maximum.at(hi, cv, uv[uv_tri.reshape(-1)] * (T - 1))
    order = np.argsort(lo[:, 0])
    for k, i in enumerate(order):                                            # sweep: compare with boxes that start before hi_x
        for j in order[k + 1:]:
            if lo[j, 0] > hi[i, 0] + 1:
                break
            assert lo[j, 1] > hi[i, 1] + 1 or lo[i, 1] > hi[j, 1] + 1, (i, j)
    # no double coverage: rasterised texels of the whole atlas against the sum of the triangle areas
    fi, _ = tex_ref.rasterize(uvatlas.uv_clip(uv), uv_tri, T, T)
    covered, expect = int((fi > 0).sum()), float(area.sum()) * (T - 1) ** 2
    assert abs(covered - expect) <= 0.02 * expect + 4 * np.sqrt(expect), (covered, expect)
    return uv, uv_tri, chart, covered


def test_chart_atlas_on_a_sphere_has_six_charts():
    import tex_support as ts
    v, f = ts.icosphere(3)
    uv, uv_tri, chart, covered = _check_atlas(v, f, 512, 1.35)
    assert int(chart.max()) + 1 == 6
    assert covered > 0.45 * 512 * 512                         # the per-face atlas wastes half of every cell on margins


def test_chart_atlas_splits_stacked_layers():
    """a helicoid of 2.5 turns: one edge-connected sheet whose normals all point up -- projected along z it covers the same
    annulus 2.5 times.  The height-field rule must cut it so that no texel is claimed twice."""
    v, f = _helicoid()
    uv, uv_tri, chart, covered = _check_atlas(v, f, 512, 1.2)
    assert int(chart.max()) + 1 >= 3


def test_chart_atlas_on_a_marching_cubes_surface_and_determinism():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import mc as omc
    ax = np.linspace(-1, 1, 48)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    vol = np.maximum(0.55 - np.sqrt((X - 0.2) ** 2 + Y ** 2 + Z ** 2), 0.4 - np.sqrt((X + 0.35) ** 2 + (Y - 0.1) ** 2 + Z ** 2))
    v, f = omc.marching_cubes(vol.astype(np.float32), 0.0)
    uv, uv_tri, chart, covered = _check_atlas(v.astype(np.float32), f, 1024, 1.4)
    assert int(chart.max()) + 1 < 0.02 * len(f)               # hundreds of faces per chart, not one
    again = uvatlas.chart_atlas(v.astype(np.float32), f, 1024)
    assert np.array_equal(again[0], uv) and np.array_equal(again[1], uv_tri)
    e = uvatlas.chart_atlas(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), 64)
    assert e[0].shape == (0, 2) and e[1].shape == (0, 3)

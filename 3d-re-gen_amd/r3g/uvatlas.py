"""UV unwraps of the texture stage (host side).  Upstream unwraps with xatlas (`hy3dgen/texgen/utils/uv_warp_utils.py:
mesh_uv_wrap`), which is not available here.

chart_atlas (round 3, the default): charts = edge-connected patches of faces whose normals share a dominant axis (+x, -x, ...,
-z), each projected orthographically along that axis (every triangle keeps its orientation, so a patch is locally injective),
patches that stack several layers over one projected spot are split by depth until they are height fields, all charts get the
SAME texel density and are shelf-packed into the texture.  A vertex is shared inside a chart and duplicated only along seams.
face_atlas (rounds 1-2, kept): every face its own chart, a right isosceles triangle in one half of a square cell of a regular
grid; vertices split per face corner (uv index = 3 * face + corner).  In both, the inpainting step fills the margins so
bilinear sampling stays inside a chart's own colours."""
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


def _face_components(faces, n_verts, labels):
    """edge-connected components of faces that carry the same label -> component id per face (0 .. n-1)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    f = np.asarray(faces, np.int64)
    F = len(f)
    e = np.sort(np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 1).reshape(-1, 2), axis=1)
    fid = np.repeat(np.arange(F), 3)
    ok = e[:, 0] != e[:, 1]
    key, fid = e[ok, 0] * np.int64(n_verts) + e[ok, 1], fid[ok]
    order = np.argsort(key, kind="stable")
    key, fid = key[order], fid[order]
    same = (key[1:] == key[:-1]) & (labels[fid[1:]] == labels[fid[:-1]])
    rows, cols = fid[:-1][same], fid[1:][same]
    _, comp = connected_components(coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(F, F)), directed=False)
    return comp


def _project(verts, faces, axis, negative):
    """orthographic projection along `axis`: (u, w) = the two other coordinates in cyclic order, u mirrored for a negative
    axis so that every triangle of the patch keeps a positive orientation; depth = the coordinate along the axis"""
    b, c = (axis + 1) % 3, (axis + 2) % 3
    u = verts[:, b] * (-1.0 if negative else 1.0)
    return u, verts[:, c], verts[:, axis]


def chart_atlas(verts, faces, tex_size, pad=2.0, max_split_rounds=6):
    """-> uv float32 [Vuv, 2] in [0, 1] (v = 0 is the top row), uv_tri int32 [F, 3], uv_to_pos int32 [Vuv] (the mesh vertex
    behind every UV vertex), chart int32 [F].  Deterministic (a pure function of the arrays)."""
    v = np.asarray(verts, np.float64).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    F, V, T = len(f), len(v), int(tex_size)
    if F == 0:
        return np.zeros((0, 2), np.float32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    axis = np.argmax(np.abs(n), axis=1)
    neg = n[np.arange(F), axis] < 0
    cls = (2 * axis + neg).astype(np.int64)
    split = np.zeros(F, np.int64)
    cen = v[f].mean(axis=1)
    # typical edge length sets the cell size of the layer test
    el = np.linalg.norm(v[f[:, 1]] - v[f[:, 0]], axis=1)
    g = max(float(np.median(el)), 1e-9)
    ar = np.arange(F)
    cu = np.floor(cen[ar, (axis + 1) % 3] / g).astype(np.int64)
    cw = np.floor(cen[ar, (axis + 2) % 3] / g).astype(np.int64)
    depth = cen[ar, axis]
    layer = np.floor(depth / (3.0 * g)).astype(np.int64)
    layer -= layer.min()
    for _ in range(max_split_rounds):
        _, label = np.unique(np.stack([cls, split], 1), axis=0, return_inverse=True)
        comp = _face_components(f, V, label.reshape(-1))
        # a patch must be a height field over its projection plane: inside one cell of the projected grid all of its faces lie
        # within three cells of depth of each other (a patch's slope is at most tan 54.7 deg = 1.41); a patch where they do not is cut into depth slabs of that thickness (two
        # sheets more than a slab apart can never share one, so one cut per patch is enough)
        _, inv = np.unique(np.stack([comp, cu, cw], 1), axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        dmin = np.full(inv.max() + 1, np.inf)
        dmax = np.full(inv.max() + 1, -np.inf)
        np.minimum.at(dmin, inv, depth)
        np.maximum.at(dmax, inv, depth)
        bad_comp = np.zeros(comp.max() + 1, bool)
        bad_comp[comp[((dmax - dmin) > 3.0 * g)[inv]]] = True
        hit = bad_comp[comp]
        if not hit.any():
            break
        split = np.where(hit, split.max() + 1 + layer, split)
    else:
        # the rounds ran out with patches still stacking layers: `comp` predates the last cut, and nothing has looked at the
        # patches that cut produced.  The per-face atlas has no such patches (the caller falls back to it on this error)
        raise ValueError("chart_atlas: patches still stack several layers over one projected spot after %d cuts"
                         % max_split_rounds)
    chart = comp.astype(np.int64)
    n_chart = int(chart.max()) + 1
    # UV vertices: one per (chart, mesh vertex) pair
    key = (chart[:, None] * np.int64(V) + f).reshape(-1)
    uniq, inv = np.unique(key, return_inverse=True)
    uv_tri = inv.reshape(F, 3).astype(np.int32)
    uv_to_pos = (uniq % V).astype(np.int32)
    uv_chart = (uniq // V).astype(np.int64)
    chart_axis = np.zeros(n_chart, np.int64)
    chart_neg = np.zeros(n_chart, bool)
    chart_axis[chart] = axis
    chart_neg[chart] = neg
    p = v[uv_to_pos]
    ax = chart_axis[uv_chart]
    pu = p[np.arange(len(p)), (ax + 1) % 3] * np.where(chart_neg[uv_chart], -1.0, 1.0)
    pw = p[np.arange(len(p)), (ax + 2) % 3]
    lo_u = np.full(n_chart, np.inf); lo_w = np.full(n_chart, np.inf)
    hi_u = np.full(n_chart, -np.inf); hi_w = np.full(n_chart, -np.inf)
    np.minimum.at(lo_u, uv_chart, pu); np.minimum.at(lo_w, uv_chart, pw)
    np.maximum.at(hi_u, uv_chart, pu); np.maximum.at(hi_w, uv_chart, pw)
    wu, ww = hi_u - lo_u, hi_w - lo_w
    order = np.lexsort((np.arange(n_chart), -ww))      # tallest first, ties by chart id: deterministic

    def pack(scale):
        """shelf packing of the charts' texel boxes; -> (x0, y0) per chart or None when the texture is too small"""
        bw = np.ceil(wu * scale).astype(np.int64) + 1 + 2 * int(np.ceil(pad))
        bh = np.ceil(ww * scale).astype(np.int64) + 1 + 2 * int(np.ceil(pad))
        if bw.max() > T or bh.max() > T:
            return None
        x0 = np.zeros(n_chart, np.int64)
        y0 = np.zeros(n_chart, np.int64)
        x = y = shelf = 0
        for c in order:
            if x + bw[c] > T:
                x, y, shelf = 0, y + shelf, 0
            if y + bh[c] > T:
                return None
            x0[c], y0[c] = x, y
            x += bw[c]
            shelf = max(shelf, bh[c])
        return x0, y0

    area = float((wu * ww).sum())
    hi_s = (T / max(np.sqrt(max(area, 1e-30)), 1e-15))          # cannot do better than 100 % fill
    lo_s = 0.0
    best = None
    s = hi_s
    for _ in range(40):                                          # find a scale that packs, then bisect upwards
        got = pack(s)
        if got is not None:
            best, lo_s = (s, got), s
            break
        s *= 0.7
    if best is None:
        raise ValueError("chart_atlas: %d charts do not fit a %d^2 texture" % (n_chart, T))
    for _ in range(10):
        mid = 0.5 * (lo_s + hi_s)
        got = pack(mid)
        if got is not None:
            best, lo_s = (mid, got), mid
        else:
            hi_s = mid
    scale, (x0, y0) = best
    m = float(np.ceil(pad))
    tu = x0[uv_chart] + m + (pu - lo_u[uv_chart]) * scale
    tw = y0[uv_chart] + m + (pw - lo_w[uv_chart]) * scale
    uv = np.stack([tu / (T - 1), tw / (T - 1)], 1).astype(np.float32)
    return uv, uv_tri, uv_to_pos, chart.astype(np.int32)

"""numpy restatement of the texture-stage primitives (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product never does).

What it restates: the behaviour of upstream Hunyuan3D-2's native helpers behind `Hunyuan3DPaintPipeline` (reference call
site src/2d_to_3d_models/run.py:97; built at :126-128) -- `custom_rasterizer.rasterize / interpolate` (CUDA), the
back-projection / baking of `hy3dgen/texgen/differentiable_renderer/mesh_render.py` and `mesh_processor.meshVerticeInpaint`
(C++).  PARITY UNPINNED: the Hunyuan3D-2 submodule is empty in /root/reference and the reference holds no fixture for this
path, so the conventions below (pixel centres, z packing, weights) are this project's statement of a standard z-buffer
rasteriser / texture baker; include/r3g.h "texture stage" documents the same rules and the HIP kernels are compared with
THIS file.  All arithmetic is float32, operation by operation in the order the kernels use (they are built without FMA
contraction), so integer outputs are compared exactly and float outputs to the last bit except where powf is involved."""
import numpy as np

F32 = np.float32
EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def _screen(pos4, tri, H, W):
    p = pos4.astype(F32)[tri]                                  # [F, 3, 4]
    w = p[..., 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (p[..., 0] / w * F32(0.5) + F32(0.5)) * F32(W - 1) + F32(0.5)
        y = (p[..., 1] / w * F32(0.5) + F32(0.5)) * F32(H - 1) + F32(0.5)
        z = p[..., 2] / w * F32(0.49999) + F32(0.5)
    area = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
    ok = (w > 0).all(axis=1) & (area != 0) & np.isfinite(area)
    return x, y, z, w, area, ok


def _bary(x, y, area, px, py):
    b0 = ((x[1] - px) * (y[2] - py) - (x[2] - px) * (y[1] - py)) / area
    b1 = ((x[2] - px) * (y[0] - py) - (x[0] - px) * (y[2] - py)) / area
    b2 = F32(1.0) - b0 - b1
    return b0, b1, b2


def rasterize(pos4, tri, H, W):
    """-> findices int32 [H, W] (face + 1, 0 empty), bary float32 [H, W, 3]"""
    tri = np.asarray(tri, np.int64).reshape(-1, 3)
    zbuf = np.full(H * W, EMPTY, np.uint64)
    if len(tri):
        x, y, z, w, area, ok = _screen(np.asarray(pos4, F32), tri, H, W)
        for f in np.nonzero(ok)[0]:
            xs, ys = x[f], y[f]
            minx, maxx, miny, maxy = xs.min(), xs.max(), ys.min(), ys.max()
            if not (maxx >= 0 and maxy >= 0 and minx <= W and miny <= H):
                continue
            ix0, ix1 = max(0, int(np.floor(minx)) - 1), min(W - 1, int(np.floor(maxx)) + 1)
            iy0, iy1 = max(0, int(np.floor(miny)) - 1), min(H - 1, int(np.floor(maxy)) + 1)
            if ix1 < ix0 or iy1 < iy0:
                continue
            gx, gy = np.meshgrid(np.arange(ix0, ix1 + 1), np.arange(iy0, iy1 + 1))
            px, py = gx.astype(F32) + F32(0.5), gy.astype(F32) + F32(0.5)
            with np.errstate(all="ignore"):
                b0, b1, b2 = _bary(xs, ys, area[f], px, py)
                depth = b0 * z[f, 0] + b1 * z[f, 1] + b2 * z[f, 2]
            inside = (b0 >= 0) & (b1 >= 0) & (b2 >= 0) & (depth >= 0) & (depth <= 1)
            if not inside.any():
                continue
            tok = (depth[inside].view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64(f + 1)
            idx = (gy[inside] * W + gx[inside]).astype(np.int64)
            np.minimum.at(zbuf, idx, tok)
    findices = np.where(zbuf == EMPTY, 0, (zbuf & np.uint64(0xFFFFFFFF))).astype(np.int32).reshape(H, W)
    bary = np.zeros((H, W, 3), F32)
    if len(tri):
        ys_, xs_ = np.nonzero(findices)
        f = findices[ys_, xs_].astype(np.int64) - 1
        px, py = xs_.astype(F32) + F32(0.5), ys_.astype(F32) + F32(0.5)
        with np.errstate(all="ignore"):
            b0, b1, b2 = _bary(x[f].T, y[f].T, area[f], px, py)
            c0, c1, c2 = b0 / w[f, 0], b1 / w[f, 1], b2 / w[f, 2]
            s = c0 + c1 + c2
            bary[ys_, xs_, 0], bary[ys_, xs_, 1], bary[ys_, xs_, 2] = c0 / s, c1 / s, c2 / s
    return findices, bary


def interpolate(attr, tri, findices, bary):
    attr = np.asarray(attr, F32)
    tri = np.asarray(tri, np.int64).reshape(-1, 3)
    shape = findices.shape
    fi = findices.reshape(-1).astype(np.int64)
    b = bary.reshape(-1, 3).astype(F32)
    out = np.zeros((len(fi), attr.shape[1]), F32)
    m = fi > 0
    t = tri[fi[m] - 1]
    out[m] = b[m, 0:1] * attr[t[:, 0]] + b[m, 1:2] * attr[t[:, 1]] + b[m, 2:3] * attr[t[:, 2]]
    return out.reshape(shape + (attr.shape[1],))


def view_weight(findices, depth, normal, cos_threshold, depth_edge, view_w, power):
    H, W = findices.shape
    n = normal.astype(F32)
    length = np.sqrt(n[..., 0] * n[..., 0] + n[..., 1] * n[..., 1] + n[..., 2] * n[..., 2])
    with np.errstate(all="ignore"):
        cosv = np.where(length > 0, n[..., 2] / length, F32(0)).astype(F32)
    keep = (findices > 0) & (cosv >= F32(cos_threshold))
    d = depth.astype(F32)
    for dx, dy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        nb_f = np.zeros_like(findices)
        nb_d = np.zeros_like(d)
        ys = slice(max(0, -dy), H - max(0, dy))
        xs = slice(max(0, -dx), W - max(0, dx))
        ys2 = slice(max(0, dy), H - max(0, -dy))
        xs2 = slice(max(0, dx), W - max(0, -dx))
        nb_f[ys, xs] = findices[ys2, xs2]
        nb_d[ys, xs] = d[ys2, xs2]
        inb = np.zeros((H, W), bool)
        inb[ys, xs] = True
        keep &= inb & (nb_f > 0) & ~(np.abs(nb_d - d) > F32(depth_edge))
    with np.errstate(all="ignore"):
        wgt = F32(view_w) * np.power(cosv, F32(power), dtype=F32)
    return np.where(keep, wgt, F32(0)).astype(F32)


def _texel(u, T):
    t = (u * F32(T - 1) + F32(0.5)).astype(np.int64)       # truncation, like the C cast
    return np.clip(t, 0, T - 1)


def _q16(c):
    return (np.clip(c, F32(0), F32(1)) * F32(65536) + F32(0.5)).astype(np.uint64)


def bake(image, weight, findices, bary, uv, uv_tri, T, acc=None):
    """accumulates into acc uint64 [T, T, 4] (created when None) and returns it"""
    acc = np.zeros((T, T, 4), np.uint64) if acc is None else acc
    uv = np.asarray(uv, F32)
    uv_tri = np.asarray(uv_tri, np.int64).reshape(-1, 3)
    fi = findices.reshape(-1).astype(np.int64)
    w = weight.reshape(-1).astype(F32)
    m = (fi > 0) & (w > 0)
    wq = (np.minimum(w[m], F32(65535)) * F32(65536) + F32(0.5)).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    nz = wq > 0
    b = bary.reshape(-1, 3).astype(F32)[m][nz]
    t = uv_tri[fi[m][nz] - 1]
    wq = wq[nz]
    u = b[:, 0] * uv[t[:, 0], 0] + b[:, 1] * uv[t[:, 1], 0] + b[:, 2] * uv[t[:, 2], 0]
    v = b[:, 0] * uv[t[:, 0], 1] + b[:, 1] * uv[t[:, 1], 1] + b[:, 2] * uv[t[:, 2], 1]
    tex = _texel(v, T) * T + _texel(u, T)
    img = image.reshape(-1, 3).astype(F32)[m][nz]
    flat = acc.reshape(-1, 4)
    for c in range(3):
        np.add.at(flat[:, c], tex, wq * _q16(img[:, c]))
    np.add.at(flat[:, 3], tex, wq)
    return acc


def bake_gather(findices_uv, bary_uv, clip_uv, uv_tri, image, weight, findices, depth, depth_eps, acc=None):
    """texel-centric baking of one view (see include/r3g.h bake_gather); accumulates into acc uint64 [T, T, 4]"""
    T = findices_uv.shape[0]
    H, W = findices.shape
    acc = np.zeros((T, T, 4), np.uint64) if acc is None else acc
    flat = acc.reshape(-1, 4)
    fu = findices_uv.reshape(-1).astype(np.int64)
    idx = np.nonzero(fu > 0)[0]
    if not len(idx):
        return acc
    t = np.asarray(uv_tri, np.int64).reshape(-1, 3)[fu[idx] - 1]
    b = bary_uv.reshape(-1, 3).astype(F32)[idx]
    c = np.asarray(clip_uv, F32)
    p = b[:, 0:1] * c[t[:, 0]] + b[:, 1:2] * c[t[:, 1]] + b[:, 2:3] * c[t[:, 2]]
    with np.errstate(all="ignore"):
        sx = (p[:, 0] / p[:, 3] * F32(0.5) + F32(0.5)) * F32(W - 1) + F32(0.5)
        sy = (p[:, 1] / p[:, 3] * F32(0.5) + F32(0.5)) * F32(H - 1) + F32(0.5)
        z = p[:, 2] / p[:, 3]
    px, py = np.floor(sx), np.floor(sy)
    ok = (p[:, 3] > 0) & (px >= 0) & (py >= 0) & (px < W) & (py < H)
    idx, sx, sy, z = idx[ok], sx[ok], sy[ok], z[ok]
    pix = py[ok].astype(np.int64) * W + px[ok].astype(np.int64)
    w = weight.reshape(-1).astype(F32)[pix]
    vis = (findices.reshape(-1)[pix] > 0) & (w > 0) & ~(z > depth.reshape(-1).astype(F32)[pix] + F32(depth_eps))
    idx, sx, sy, w = idx[vis], sx[vis], sy[vis], w[vis]
    wq = (np.minimum(w, F32(65535)) * F32(65536) + F32(0.5)).astype(np.uint64)
    nz = wq > 0
    idx, sx, sy, wq = idx[nz], sx[nz], sy[nz], wq[nz]
    fx, fy = sx - F32(0.5), sy - F32(0.5)
    x0f, y0f = np.floor(fx), np.floor(fy)
    ax, ay = (fx - x0f).astype(F32), (fy - y0f).astype(F32)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    x1, y1 = np.clip(x0 + 1, 0, W - 1), np.clip(y0 + 1, 0, H - 1)
    x0, y0 = np.clip(x0, 0, W - 1), np.clip(y0, 0, H - 1)
    img = image.astype(F32)
    for ch in range(3):
        c00, c01, c10, c11 = img[y0, x0, ch], img[y0, x1, ch], img[y1, x0, ch], img[y1, x1, ch]
        top = c00 + (c01 - c00) * ax
        bot = c10 + (c11 - c10) * ax
        flat[idx, ch] += wq * _q16(top + (bot - top) * ay)
    flat[idx, 3] += wq
    return acc


def bake_finalize(acc):
    w = acc[..., 3]
    with np.errstate(all="ignore"):
        tex = np.where(w[..., None] > 0, acc[..., :3].astype(np.float64) / (w[..., None].astype(np.float64) * 65536.0), 0.0)
    return tex.astype(F32), (w > 0).astype(np.uint8)


def inpaint(tex, mask, findices_uv, bary_uv, verts, pos_tri, uv, uv_tri, dilate_iters, max_rounds=8192):
    """-> texture float32 [T, T, 3], mask uint8 [T, T] (1 painted, 2 from vertex colours, 3 dilated), propagation rounds"""
    T = tex.shape[0]
    tex = tex.astype(F32).copy().reshape(-1, 3)
    mask = mask.astype(np.uint8).copy().reshape(-1)
    verts = np.asarray(verts, F32)
    uv = np.asarray(uv, F32)
    pos_c = np.asarray(pos_tri, np.int64).reshape(-1)
    uv_c = np.asarray(uv_tri, np.int64).reshape(-1)
    V = len(verts)
    # the texel a corner takes its vertex colour from: the corner's UV pulled a quarter of the way to the chart triangle's centroid
    f3 = (np.arange(len(uv_c)) // 3) * 3
    j0, j1, j2 = uv_c[f3], uv_c[f3 + 1], uv_c[f3 + 2]
    third = F32(1.0) / F32(3.0)
    cu = ((uv[j0, 0] + uv[j1, 0]) + uv[j2, 0]) * third
    cv = ((uv[j0, 1] + uv[j1, 1]) + uv[j2, 1]) * third
    pu = uv[uv_c, 0] * F32(0.75) + cu * F32(0.25)
    pv = uv[uv_c, 1] * F32(0.75) + cv * F32(0.25)
    corner_tex = _texel(pv.astype(F32), T) * T + _texel(pu.astype(F32), T)
    owner = np.full(V, 0xFFFFFFFF, np.uint64)
    painted = mask[corner_tex] > 0
    np.minimum.at(owner, pos_c[painted], np.nonzero(painted)[0].astype(np.uint64))
    vmask = owner != 0xFFFFFFFF
    vcolor = np.zeros((V, 3), F32)
    vcolor[vmask] = tex[corner_tex[owner[vmask].astype(np.int64)]]
    # propagation: directed edges a <- b inside every face
    a_all = np.concatenate([pos_c, pos_c])
    f3 = (np.arange(len(pos_c)) // 3) * 3
    k = np.arange(len(pos_c)) % 3
    b_all = np.concatenate([pos_c[f3 + (k + 1) % 3], pos_c[f3 + (k + 2) % 3]])
    d = verts[a_all] - verts[b_all]
    w = F32(1.0) / (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2] + F32(1e-6))
    wq_all = (w * F32(1024) + F32(0.5)).astype(np.uint64)
    rounds = 0
    while rounds < max_rounds:
        sel = vmask[b_all] & ~vmask[a_all] & (a_all != b_all) & (wq_all > 0)
        acc = np.zeros((V, 4), np.uint64)
        if sel.any():
            aa, bb, wq = a_all[sel], b_all[sel], wq_all[sel]
            for c in range(3):
                np.add.at(acc[:, c], aa, wq * _q16(vcolor[bb, c]))
            np.add.at(acc[:, 3], aa, wq)
        new = ~vmask & (acc[:, 3] > 0)
        if not new.any():
            break
        vcolor[new] = (acc[new, :3].astype(np.float64) / (acc[new, 3:4].astype(np.float64) * 65536.0)).astype(F32)
        vmask = vmask | new
        rounds += 1
    # unpainted covered texels from the vertex colours
    fi = findices_uv.reshape(-1).astype(np.int64)
    b = bary_uv.reshape(-1, 3).astype(F32)
    tri = np.asarray(pos_tri, np.int64).reshape(-1, 3)
    cand = np.nonzero((mask == 0) & (fi > 0))[0]
    if len(cand):
        t = tri[fi[cand] - 1]
        s = np.zeros(len(cand), F32)
        rgb = np.zeros((len(cand), 3), F32)
        for kk in range(3):
            use = vmask[t[:, kk]]
            wk = np.where(use, b[cand, kk], F32(0)).astype(F32)
            s = np.where(use, s + wk, s).astype(F32)
            rgb = np.where(use[:, None], rgb + wk[:, None] * vcolor[t[:, kk]], rgb).astype(F32)
        good = s > 0
        tex[cand[good]] = rgb[good] / s[good, None]
        mask[cand[good]] = 2
    # dilation (Jacobi steps, neighbours in row-major order)
    tex = tex.reshape(T, T, 3)
    mask = mask.reshape(T, T)
    for _ in range(dilate_iters):
        s = np.zeros((T, T), F32)
        a = np.zeros((T, T, 3), F32)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0:
                    continue
                ys = slice(max(0, -dy), T - max(0, dy))
                xs = slice(max(0, -dx), T - max(0, dx))
                ys2 = slice(max(0, dy), T - max(0, -dy))
                xs2 = slice(max(0, dx), T - max(0, -dx))
                nb = np.zeros((T, T), bool)
                nb[ys, xs] = mask[ys2, xs2] > 0
                col = np.zeros((T, T, 3), F32)
                col[ys, xs] = tex[ys2, xs2]
                s = np.where(nb, s + F32(1), s).astype(F32)
                a = np.where(nb[..., None], a + col, a).astype(F32)
        grow = (mask == 0) & (s > 0)
        with np.errstate(all="ignore"):
            tex = np.where(grow[..., None], a / s[..., None], tex).astype(F32)
        mask = np.where(grow, 3, mask).astype(np.uint8)
    return tex, mask, rounds

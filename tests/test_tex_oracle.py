"""Sanity of the numpy restatement of the texture-stage primitives (oracle/tex_ref.py): it is parity-unpinned (no upstream
source or fixture exists here), so these tests pin it to properties any z-buffer rasteriser / texture baker must have."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import tex_ref  # noqa: E402
import tex_support as ts  # noqa: E402


def test_single_triangle_coverage_and_barycentrics():
    H = W = 64
    pos = np.array([[-0.8, -0.8, 0, 1], [0.8, -0.8, 0, 1], [-0.8, 0.8, 0, 1]], np.float32)
    fi, bary = tex_ref.rasterize(pos, np.array([[0, 1, 2]], np.int32), H, W)
    covered = fi > 0
    area_px = 0.5 * (1.6 / 2 * (W - 1)) * (1.6 / 2 * (H - 1))
    assert abs(covered.sum() - area_px) < 0.06 * area_px
    assert np.allclose(bary[covered].sum(axis=1), 1.0, atol=1e-5) and (bary[covered] >= 0).all()
    # interpolating the screen positions of the corners gives back the pixel centre
    x = (pos[:, 0] * 0.5 + 0.5) * (W - 1) + 0.5
    y = (pos[:, 1] * 0.5 + 0.5) * (H - 1) + 0.5
    out = tex_ref.interpolate(np.stack([x, y], 1), np.array([[0, 1, 2]]), fi, bary)
    ys, xs = np.nonzero(covered)
    assert np.allclose(out[ys, xs, 0], xs + 0.5, atol=1e-3) and np.allclose(out[ys, xs, 1], ys + 0.5, atol=1e-3)


def test_nearest_wins_and_ties_go_to_the_smaller_face():
    H = W = 32
    quad = lambda z: [[-0.9, -0.9, z, 1], [0.9, -0.9, z, 1], [0.9, 0.9, z, 1]]   # noqa: E731
    pos = np.array(quad(0.5) + quad(-0.5) + quad(-0.5), np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    fi, _ = tex_ref.rasterize(pos, tri, H, W)
    assert set(np.unique(fi)) == {0, 2}            # face 1 (z = -0.5, nearer) hides face 0; its duplicate (face 2) loses the tie


def test_perspective_correct_interpolation():
    H = W = 48
    # a quad receding in depth: attribute = world x; perspective-correct interpolation is linear in world space
    pos_w = np.array([[-1, -1, 1.0], [1, -1, 1.0], [1, 1, 3.0], [-1, 1, 3.0]], np.float32)
    w = pos_w[:, 2:3]
    pos = np.concatenate([pos_w[:, :2], np.zeros((4, 1), np.float32), w], 1).astype(np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    fi, bary = tex_ref.rasterize(pos, tri, H, W)
    attr = tex_ref.interpolate(pos_w[:, :1] * 1.0, tri, fi, bary)[..., 0]
    wi = tex_ref.interpolate(w, tri, fi, bary)[..., 0]
    ys, xs = np.nonzero(fi)
    ndc_x = ((xs + 0.5) - 0.5) / (W - 1) * 2 - 1
    assert np.allclose(attr[ys, xs], ndc_x * wi[ys, xs], atol=2e-3)


def test_bake_recovers_a_constant_and_inpaint_completes_a_sphere():
    v, f = ts.icosphere(2)
    T, R = 128, 96
    uv, uv_tri = ts.face_atlas(f, T)
    acc = None
    normals = v / np.linalg.norm(v, axis=1, keepdims=True)
    for deg in (0,):
        rot = ts.rot_y(deg)
        clip, cam = ts.ortho_clip(v, rot)
        fi, bary = tex_ref.rasterize(clip, f, R, R)
        depth = tex_ref.interpolate(clip[:, 2:3], f, fi, bary)[..., 0]
        nrm = tex_ref.interpolate(normals @ rot.T, f, fi, bary)
        wgt = tex_ref.view_weight(fi, depth, nrm, 0.2, 0.05, 1.0, 2.0)
        assert (wgt[fi == 0] == 0).all() and wgt.max() > 0.5
        img = np.zeros((R, R, 3), np.float32)
        img[...] = (0.25, 0.5, 0.75)
        acc = tex_ref.bake(img, wgt, fi, bary, uv, uv_tri, T, acc)
    tex, mask = tex_ref.bake_finalize(acc)
    assert mask.sum() > 0 and np.allclose(tex[mask > 0], (0.25, 0.5, 0.75), atol=2e-5)
    uvc = np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32), np.ones((len(uv), 1), np.float32)], 1)
    fi_uv, bary_uv = tex_ref.rasterize(uvc, uv_tri, T, T)
    tex2, mask2, rounds = tex_ref.inpaint(tex, mask, fi_uv, bary_uv, v, f, uv, uv_tri, dilate_iters=2)
    assert rounds >= 1
    assert ((mask2 > 0) | (fi_uv == 0)).all() or (mask2[fi_uv > 0] > 0).all()    # every covered texel has a colour
    assert np.allclose(tex2[mask2 > 0], (0.25, 0.5, 0.75), atol=1e-4)              # a constant stays a constant
    assert (mask2 == 3).sum() > 0 and ((mask2 > 0).sum() > (fi_uv > 0).sum())      # the dilation grew into the gutter


def test_vertex_colours_are_seeded_inside_the_chart():
    """the texel nearest to a chart CORNER usually lies outside the triangle (unpainted), so most corners gave no
    seed.  With the seed pulled a quarter of the way to the chart's centroid, a texture that is painted on every covered
    texel seeds every vertex directly: no propagation round is needed, and a vertex's colour is its own chart's colour."""
    v, f = ts.icosphere(2)
    T = 256
    uv, uv_tri = ts.face_atlas(f, T)
    uvc = np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32), np.ones((len(uv), 1), np.float32)], 1)
    fi_uv, bary_uv = tex_ref.rasterize(uvc, uv_tri, T, T)
    covered = fi_uv > 0
    rng = np.random.default_rng(0)
    face_colour = rng.uniform(0.1, 0.9, (len(f), 3)).astype(np.float32)
    tex = np.zeros((T, T, 3), np.float32)
    tex[covered] = face_colour[fi_uv[covered] - 1]
    mask = covered.astype(np.uint8)
    tex2, mask2, rounds = tex_ref.inpaint(tex, mask, fi_uv, bary_uv, v, f, uv, uv_tri, dilate_iters=0)
    assert rounds == 0                                    # every vertex found a painted texel inside one of its charts
    assert np.array_equal(tex2[covered], tex[covered])    # painted texels are never touched
    # half of the corners' own nearest texels are NOT covered: that is what the old rule looked at
    own = (np.clip((uv[:, 1] * (T - 1) + 0.5).astype(int), 0, T - 1), np.clip((uv[:, 0] * (T - 1) + 0.5).astype(int), 0, T - 1))
    assert (~covered[own]).mean() > 0.3

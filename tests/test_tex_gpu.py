"""Texture-stage HIP kernels (C ABI r3g_tex_*) against the numpy restatement oracle/tex_ref.py on the same seeded inputs:
integer outputs (face ids, masks, accumulators) bit-exact, float outputs bit-exact except where powf enters."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu


def _t(a, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize("nv,nf,hw,persp,seed", [(40, 60, (64, 64), True, 1), (300, 900, (97, 131), True, 2),
                                                 (50, 80, (33, 200), False, 3), (3, 3, (16, 16), False, 4)])
def test_rasterize_and_interpolate_match_the_oracle(nv, nf, hw, persp, seed):
    from oracle import tex_ref
    from r3g import texops
    import tex_support as ts
    H, W = hw
    pos, tri = ts.random_soup(nv, max(nf, 3), seed, persp)
    fi_ref, bary_ref = tex_ref.rasterize(pos, tri, H, W)
    fi, bary = texops.rasterize(_t(pos), _t(tri), H, W)
    assert np.array_equal(fi.cpu().numpy(), fi_ref)
    assert np.array_equal(bary.cpu().numpy(), bary_ref)
    assert (fi_ref > 0).any()
    attr = np.random.default_rng(seed).normal(size=(nv, 5)).astype(np.float32)
    out = texops.interpolate(_t(attr), _t(tri), fi, bary).cpu().numpy()
    assert np.array_equal(out, tex_ref.interpolate(attr, tri, fi_ref, bary_ref))


def test_empty_mesh_and_offscreen_mesh():
    import torch
    from r3g import texops
    fi, bary = texops.rasterize(torch.zeros((0, 4), device="cuda"), torch.zeros((0, 3), dtype=torch.int32, device="cuda"), 8, 9)
    assert fi.shape == (8, 9) and int(fi.abs().sum()) == 0 and float(bary.abs().sum()) == 0.0
    pos = np.array([[5, 5, 0, 1], [6, 5, 0, 1], [5, 6, 0, 1]], np.float32)
    fi, _ = texops.rasterize(_t(pos), _t(np.array([[0, 1, 2]], np.int32)), 16, 16)
    assert int(fi.sum()) == 0


def _sphere_scene(T, R, views):
    from oracle import tex_ref
    import tex_support as ts
    v, f = ts.icosphere(3)
    uv, uv_tri = ts.face_atlas(f, T)
    normals = v / np.linalg.norm(v, axis=1, keepdims=True)
    rng = np.random.default_rng(7)
    scene = []
    for deg in views:
        rot = ts.rot_y(deg)
        clip, _ = ts.ortho_clip(v, rot)
        img = rng.random((R, R, 3), dtype=np.float32)
        scene.append((clip, (normals @ rot.T).astype(np.float32), img))
    return v, f, uv, uv_tri, scene


def test_bake_two_views_and_inpaint_match_the_oracle():
    from oracle import tex_ref
    from r3g import texops
    T, R = 160, 120
    v, f, uv, uv_tri, scene = _sphere_scene(T, R, (0, 70))
    acc_ref = None
    acc = texops.new_accumulator(T, "cuda")
    for clip, nrm_v, img in scene:
        fi_ref, bary_ref = tex_ref.rasterize(clip, f, R, R)
        depth_ref = tex_ref.interpolate(clip[:, 2:3], f, fi_ref, bary_ref)[..., 0]
        nmap_ref = tex_ref.interpolate(nrm_v, f, fi_ref, bary_ref)
        w_ref = tex_ref.view_weight(fi_ref, depth_ref, nmap_ref, 0.2, 0.05, 0.8, 3.0)
        fi, bary = texops.rasterize(_t(clip), _t(f), R, R)
        depth = texops.interpolate(_t(clip[:, 2:3]), _t(f), fi, bary)[..., 0]
        nmap = texops.interpolate(_t(nrm_v), _t(f), fi, bary)
        w = texops.view_weight(fi, depth, nmap, 0.2, 0.05, 0.8, 3.0)
        wn = w.cpu().numpy()
        assert np.array_equal(wn > 0, w_ref > 0)                      # same pixels kept (threshold, silhouette, depth edges)
        assert np.allclose(wn, w_ref, rtol=2e-6, atol=0)              # powf: last-bit differences allowed
        # bake with the ORACLE's weights on both sides so that the integer accumulators must agree exactly
        acc_ref = tex_ref.bake(img, w_ref, fi_ref, bary_ref, uv, uv_tri, T, acc_ref)
        texops.bake(_t(img), _t(w_ref), fi, bary, _t(uv), _t(uv_tri), acc)
    assert np.array_equal(acc.cpu().numpy().view(np.uint64), acc_ref)
    # texel-centric baking of the same views (oracle weights on both sides: integer accumulators must agree exactly)
    uvc = np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32), np.ones((len(uv), 1), np.float32)], 1)
    fu_ref, bu_ref = tex_ref.rasterize(uvc, uv_tri, T, T)
    fu, bu = texops.rasterize(_t(uvc), _t(uv_tri), T, T)
    g_ref, g = None, texops.new_accumulator(T, "cuda")
    for clip, nrm_v, img in scene:
        fi_ref, bary_ref = tex_ref.rasterize(clip, f, R, R)
        depth_ref = tex_ref.interpolate(clip[:, 2:3], f, fi_ref, bary_ref)[..., 0]
        w_ref = tex_ref.view_weight(fi_ref, depth_ref, tex_ref.interpolate(nrm_v, f, fi_ref, bary_ref), 0.2, 0.05, 0.8, 3.0)
        clip_uv = np.ascontiguousarray(clip[f].reshape(-1, 4))
        g_ref = tex_ref.bake_gather(fu_ref, bu_ref, clip_uv, uv_tri, img, w_ref, fi_ref, depth_ref, 0.02, g_ref)
        texops.bake_gather(fu, bu, _t(clip_uv), _t(uv_tri), _t(img), _t(w_ref), _t(fi_ref), _t(depth_ref), g, 0.02)
    assert np.array_equal(g.cpu().numpy().view(np.uint64), g_ref)
    assert 0.2 < (g_ref[..., 3] > 0).sum() / (fu_ref > 0).sum() < 0.9        # two views see part of the sphere, densely
    tex_ref_, mask_ref = tex_ref.bake_finalize(acc_ref)
    tex, mask = texops.bake_finalize(acc)
    assert np.array_equal(mask.cpu().numpy(), mask_ref) and np.array_equal(tex.cpu().numpy(), tex_ref_)
    assert 0.2 < mask_ref[tex_ref.rasterize(np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32),
                                                            np.ones((len(uv), 1), np.float32)], 1), uv_tri, T, T)[0] > 0].mean() < 0.9

    uvc = np.concatenate([uv * 2 - 1, np.zeros((len(uv), 1), np.float32), np.ones((len(uv), 1), np.float32)], 1)
    fi_uv_ref, bary_uv_ref = tex_ref.rasterize(uvc, uv_tri, T, T)
    fi_uv, bary_uv = texops.rasterize(_t(uvc), _t(uv_tri), T, T)
    assert np.array_equal(fi_uv.cpu().numpy(), fi_uv_ref) and np.array_equal(bary_uv.cpu().numpy(), bary_uv_ref)
    t2_ref, m2_ref, rounds_ref = tex_ref.inpaint(tex_ref_, mask_ref, fi_uv_ref, bary_uv_ref, v, f, uv, uv_tri, dilate_iters=3)
    t2, m2, rounds = texops.inpaint(tex, mask, fi_uv, bary_uv, _t(v), _t(f), _t(uv), _t(uv_tri), dilate_iters=3)
    assert rounds == rounds_ref and rounds >= 2
    assert np.array_equal(m2.cpu().numpy(), m2_ref)
    assert np.array_equal(t2.cpu().numpy(), t2_ref)
    assert (m2_ref[fi_uv_ref > 0] > 0).all()


def test_rasterizer_at_texture_scale_is_deterministic_and_fast():
    """40 000-face mesh into 2048^2: two runs give identical images (the z-buffer does not depend on scheduling)"""
    import torch
    from r3g import texops
    import tex_support as ts
    from parity_support import report
    v, f = ts.icosphere(6)              # 81 920 faces
    clip, _ = ts.ortho_clip(v, ts.rot_y(20))
    pc, tf = _t(clip), _t(f)
    a = texops.rasterize(pc, tf, 2048, 2048)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    b = texops.rasterize(pc, tf, 2048, 2048)
    ev[1].record()
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    cover = float((a[0] > 0).float().mean())
    assert 0.4 < cover < 0.6           # a unit sphere in a 2.4-wide window: pi / 5.76 = 0.545
    report("rasterize 81 920 faces into 2048^2: milliseconds", ev[0].elapsed_time(ev[1]), 50.0)

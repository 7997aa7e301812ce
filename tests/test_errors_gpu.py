"""Error behaviour of the C ABI on the GPU box: every entry point returns a negative R3G_ERR_* code and leaves a
message in r3g_last_error() instead of crashing -- call order (emit without count, grid query without decode), bad
arguments, unknown options, missing weights."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ERR_INVALID, ERR_STATE = -1, -4


def _msg(L):
    return L.r3g_last_error().decode()


def test_marching_cubes_call_order_and_arguments():
    from r3g import ffi
    L, ctx = ffi.lib(), ffi.context(0)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    g = torch.zeros(4, 4, 4, device="cuda")
    v = torch.zeros(16, 3, device="cuda")
    f = torch.zeros(16, 3, dtype=torch.int32, device="cuda")
    # a failed count (no surface) must not leave an emit-able state behind
    assert L.r3g_mc_count(ctx, g.data_ptr(), 4, 4, 4, 0.0, 0, ctypes.byref(nv), ctypes.byref(nf), s) == ffi.R3G_ERR_NO_SURFACE
    assert L.r3g_mc_emit(ctx, v.data_ptr(), f.data_ptr(), None, 0, s) == ERR_STATE and "r3g_mc_count" in _msg(L)
    assert L.r3g_mc_count(ctx, g.data_ptr(), 1, 4, 4, 0.0, 0, ctypes.byref(nv), ctypes.byref(nf), s) == ERR_INVALID
    assert "at least 2x2x2" in _msg(L)                       # skimage's message
    assert L.r3g_mc_count(ctx, None, 4, 4, 4, 0.0, 0, ctypes.byref(nv), ctypes.byref(nf), s) == ERR_INVALID
    assert L.r3g_mc_count(ctx, g.data_ptr(), 2048, 2048, 2048, 0.0, 0, ctypes.byref(nv), ctypes.byref(nf), s) == ERR_INVALID
    assert "too large" in _msg(L)
    with pytest.raises(ffi.LevelRangeError):
        ffi.check(L.r3g_mc_count(ctx, g.data_ptr(), 4, 4, 4, 1.0, 0, ctypes.byref(nv), ctypes.byref(nf), s))


def test_mesh_and_option_arguments():
    from r3g import ffi
    L, ctx = ffi.lib(), ffi.context(0)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    v = torch.zeros(8, 3, device="cuda")
    f = torch.zeros(4, 3, dtype=torch.int32, device="cuda")
    nv, nf = ctypes.c_int64(8), ctypes.c_int64(4)
    assert L.r3g_mesh_reduce_faces(ctx, v.data_ptr(), ctypes.byref(nv), f.data_ptr(), ctypes.byref(nf), 0, s) == ERR_INVALID
    assert L.r3g_mesh_remove_degenerate(ctx, None, ctypes.byref(nv), f.data_ptr(), ctypes.byref(nf), s) == ERR_INVALID
    neg = ctypes.c_int64(-1)
    assert L.r3g_mesh_remove_floaters(ctx, v.data_ptr(), ctypes.byref(neg), f.data_ptr(), ctypes.byref(nf), 0.005, s) == ERR_INVALID
    assert L.r3g_set_option(b"no_such_switch", 1) == ERR_INVALID and "no_such_switch" in _msg(L)
    assert L.r3g_set_option(None, 1) == ERR_INVALID
    cnt = (ctypes.c_int64 * 2)()
    ms = (ctypes.c_double * 2)()
    assert L.r3g_prof_read(cnt, ms, ms, 2) == ERR_INVALID    # too few slots


def test_model_call_order_and_missing_weights():
    from oracle import hy3d_torch as H
    from r3g import ffi, model as M
    L = ffi.lib()
    cfg = H.tiny_config()
    sd = H.synthetic_state_dict(cfg, seed=1)
    m = M.ShapeModel(cfg, sd, 0)
    grid = torch.empty(9, 9, 9, device="cuda")
    with pytest.raises(ffi.R3GError) as e:                   # grid query before the VAE decode of this model
        m.grid_query(1.01, 8, grid)
    assert e.value.code == ERR_STATE and "r3g_vae_decode" in str(e.value)
    m.vae_decode(torch.randn(cfg["vae"]["num_latents"], cfg["vae"]["embed_dim"]))
    with pytest.raises(ffi.R3GError) as e:                   # range outside the grid
        m.grid_query(1.01, 8, grid, start=700, count=100)
    assert e.value.code == ERR_INVALID
    m.grid_query(1.01, 8, grid)
    assert torch.isfinite(grid).all()
    broken = dict(sd)
    del broken["model.double_blocks.0.img_attn.qkv.weight"]
    m2 = M.ShapeModel(cfg, broken, 0)
    x = torch.randn(1, cfg["vae"]["num_latents"], cfg["dit"]["in_channels"])
    cond = torch.randn(1, 26, cfg["dit"]["context_in_dim"])
    with pytest.raises(ffi.R3GError) as e:
        m2.dit_forward(x, torch.tensor([0.5]), cond)
    assert e.value.code == ERR_STATE and "img_attn.qkv" in str(e.value)
    assert np.isfinite(m.dit_forward(x, torch.tensor([0.5]), cond).cpu().numpy()).all()   # the intact model still works

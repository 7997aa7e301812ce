"""Texture-stage primitives on device buffers (include/r3g.h "texture stage"): the gfx950 counterparts of upstream's
`custom_rasterizer` (rasterize / interpolate), of the view baking in its MeshRender and of `mesh_processor.meshVerticeInpaint`.
Inputs and outputs are torch CUDA tensors (device memory only: torch does no arithmetic here); images are [H, W, C] float32
with row 0 at the top.  There is no CPU path."""
import ctypes

import torch

from . import ffi


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _locked(sync):
    """the shared context's texture workspace (z-buffer, inpainting buffers) belongs to one call at a time; `sync`: the
    call leaves kernels behind that still use it, so the stream is drained before the next caller may enter"""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapper(*args, **kw):
            first = next(a for a in list(args) + list(kw.values()) if isinstance(a, torch.Tensor))
            if not first.is_cuda:
                return fn(*args, **kw)          # the function raises the "no CPU path" error itself
            with ffi.device_lock(first.device.index or 0):
                out = fn(*args, **kw)
                if sync:
                    torch.cuda.current_stream(first.device).synchronize()
                return out
        return wrapper
    return deco


def _dev(t, dtype, what):
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (there is no CPU path)" % what)
    return t.detach().to(dtype).contiguous()


@_locked(True)
def rasterize(pos_clip, tri, height, width):
    """pos_clip float32 [V, 4] (clip space), tri int32 [F, 3] -> findices int32 [H, W] (face + 1; 0 = empty),
    bary float32 [H, W, 3] (perspective-correct)"""
    pos = _dev(pos_clip, torch.float32, "pos_clip")
    t = _dev(tri, torch.int32, "tri")
    if pos.ndim != 2 or pos.shape[1] != 4 or t.ndim != 2 or t.shape[1] != 3:
        raise ValueError("expected pos_clip [V,4] and tri [F,3]")
    fi = torch.empty((height, width), dtype=torch.int32, device=pos.device)
    bary = torch.empty((height, width, 3), dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        ffi.check(ffi.lib().r3g_tex_rasterize(ffi.context(pos.device.index or 0), _p(pos), pos.shape[0], _p(t), t.shape[0],
                                              int(height), int(width), _p(fi), _p(bary), _s()))
    return fi, bary


def interpolate(attr, tri, findices, bary):
    """attr float32 [V, C] -> [H, W, C]: sum_k bary[..., k] * attr[tri[face, k]] (zeros where findices == 0)"""
    a = _dev(attr, torch.float32, "attr")
    t = _dev(tri, torch.int32, "tri")
    fi = _dev(findices, torch.int32, "findices")
    b = _dev(bary, torch.float32, "bary")
    out = torch.empty(tuple(fi.shape) + (a.shape[1],), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        ffi.check(ffi.lib().r3g_tex_interpolate(ffi.context(a.device.index or 0), _p(a), int(a.shape[1]), _p(t), _p(fi), _p(b),
                                                fi.numel(), _p(out), _s()))
    return out


def view_weight(findices, depth, normal, cos_threshold=0.1, depth_edge=0.01, view_weight=1.0, power=4.0):
    fi = _dev(findices, torch.int32, "findices")
    d = _dev(depth, torch.float32, "depth")
    n = _dev(normal, torch.float32, "normal")
    h, w = fi.shape
    out = torch.empty((h, w), dtype=torch.float32, device=fi.device)
    with torch.cuda.device(fi.device):
        ffi.check(ffi.lib().r3g_tex_view_weight(ffi.context(fi.device.index or 0), _p(fi), _p(d), _p(n), h, w, float(cos_threshold),
                                                float(depth_edge), float(view_weight), float(power), _p(out), _s()))
    return out


def new_accumulator(tex_size, device):
    """uint64 [T, T, 4] fixed-point accumulator (carried as int64 storage), zeroed"""
    return torch.zeros((tex_size, tex_size, 4), dtype=torch.int64, device=device)


def bake(image, weight, findices, bary, uv, uv_tri, acc):
    """scatter one view into `acc` (new_accumulator); image float32 [H, W, 3] in [0, 1], weight float32 [H, W]"""
    img = _dev(image, torch.float32, "image")
    w = _dev(weight, torch.float32, "weight")
    fi = _dev(findices, torch.int32, "findices")
    b = _dev(bary, torch.float32, "bary")
    u = _dev(uv, torch.float32, "uv")
    ut = _dev(uv_tri, torch.int32, "uv_tri")
    if not (acc.is_cuda and acc.dtype == torch.int64 and acc.is_contiguous()):
        raise ValueError("acc must come from new_accumulator()")
    with torch.cuda.device(img.device):
        ffi.check(ffi.lib().r3g_tex_bake(ffi.context(img.device.index or 0), _p(img), _p(w), _p(fi), _p(b), _p(u), _p(ut),
                                         fi.numel(), int(acc.shape[0]), _p(acc), _s()))
    return acc


def bake_gather(findices_uv, bary_uv, clip_uv, uv_tri, image, weight, findices, depth, acc, depth_eps=0.01):
    """texel-centric baking of one view into `acc`: clip_uv float32 [Vuv, 4] = this view's clip coordinates of the UV
    vertices, depth float32 [H, W] = interpolate() of clip z / w over the view"""
    fu = _dev(findices_uv, torch.int32, "findices_uv")
    bu = _dev(bary_uv, torch.float32, "bary_uv")
    cu = _dev(clip_uv, torch.float32, "clip_uv")
    ut = _dev(uv_tri, torch.int32, "uv_tri")
    img = _dev(image, torch.float32, "image")
    w = _dev(weight, torch.float32, "weight")
    fi = _dev(findices, torch.int32, "findices")
    d = _dev(depth, torch.float32, "depth")
    if not (acc.is_cuda and acc.dtype == torch.int64 and acc.is_contiguous() and acc.shape[0] == fu.shape[0]):
        raise ValueError("acc must come from new_accumulator() of the UV raster's size")
    h, wd = fi.shape
    with torch.cuda.device(img.device):
        ffi.check(ffi.lib().r3g_tex_bake_gather(ffi.context(img.device.index or 0), _p(fu), _p(bu), _p(cu), _p(ut), int(fu.shape[0]),
                                                _p(img), _p(w), _p(fi), _p(d), int(h), int(wd), float(depth_eps), _p(acc), _s()))
    return acc


def bake_finalize(acc):
    """-> texture float32 [T, T, 3], mask uint8 [T, T] (1 where some view painted the texel)"""
    t = int(acc.shape[0])
    tex = torch.empty((t, t, 3), dtype=torch.float32, device=acc.device)
    mask = torch.empty((t, t), dtype=torch.uint8, device=acc.device)
    with torch.cuda.device(acc.device):
        ffi.check(ffi.lib().r3g_tex_bake_finalize(ffi.context(acc.device.index or 0), _p(acc), t, _p(tex), _p(mask), _s()))
    return tex, mask


@_locked(False)
def inpaint(texture, mask, findices_uv, bary_uv, verts, pos_tri, uv, uv_tri, dilate_iters=8):
    """-> (texture, mask, propagation rounds); mask: 1 painted by a view, 2 filled from vertex colours, 3 dilated, 0 empty"""
    tex = _dev(texture, torch.float32, "texture").clone()
    m = _dev(mask, torch.uint8, "mask").clone()
    fi = _dev(findices_uv, torch.int32, "findices_uv")
    b = _dev(bary_uv, torch.float32, "bary_uv")
    v = _dev(verts, torch.float32, "verts")
    pt = _dev(pos_tri, torch.int32, "pos_tri")
    u = _dev(uv, torch.float32, "uv")
    ut = _dev(uv_tri, torch.int32, "uv_tri")
    rounds = ctypes.c_int(0)
    with torch.cuda.device(tex.device):
        ffi.check(ffi.lib().r3g_tex_inpaint(ffi.context(tex.device.index or 0), _p(tex), _p(m), int(tex.shape[0]), _p(fi), _p(b),
                                            _p(v), v.shape[0], _p(pt), _p(u), _p(ut), pt.shape[0], int(dilate_iters),
                                            ctypes.byref(rounds), _s()))
    return tex, m, rounds.value

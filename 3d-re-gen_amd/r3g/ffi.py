"""ctypes binding of include/r3g.h.  There is no Python/CPU fallback: a missing or unloadable
libr3g.so, or a missing GPU, raises immediately."""
import ctypes
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (R3G_LIBRARY: a measurement hook -- `tools/r06_gpu.sh ablib` times two BUILDS of the library against each other on one box)
LIB_PATH = os.environ.get("R3G_LIBRARY") or os.path.join(_PKG, "libr3g.so")

R3G_ERR_LEVEL_RANGE = -10
R3G_ERR_NO_SURFACE = -11


class R3GError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libr3g error %d: %s" % (code, msg))
        self.code = code


class LevelRangeError(ValueError):
    """skimage: ValueError('Surface level must be within volume data range.')"""


class NoSurfaceError(RuntimeError):
    """skimage: RuntimeError('No surface found at the given iso value.')"""


# every symbol include/r3g.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_I64P = ctypes.POINTER(ctypes.c_int64)
SYMBOLS = {
    "r3g_version": (_I, []),
    "r3g_last_error": (ctypes.c_char_p, []),
    "r3g_create": (_I, [_I, ctypes.POINTER(_P)]),
    "r3g_destroy": (None, [_P]),
    "r3g_mc_count": (_I, [_P, _P, _I, _I, _I, _D, _I, _I64P, _I64P, _P]),
    "r3g_mc_emit": (_I, [_P, _P, _P, _P, _I, _P]),
    "r3g_mesh_remove_floaters": (_I, [_P, _P, _I64P, _P, _I64P, _D, _P]),
    "r3g_mesh_remove_degenerate": (_I, [_P, _P, _I64P, _P, _I64P, _P]),
    "r3g_mesh_reduce_faces": (_I, [_P, _P, _I64P, _P, _I64P, ctypes.c_int64, _P]),
    "r3g_mesh_cluster_faces": (_I, [_P, _P, _I64P, _P, _I64P, ctypes.c_int64, _P]),
    "r3g_tex_rasterize": (_I, [_P, _P, ctypes.c_int64, _P, ctypes.c_int64, _I, _I, _P, _P, _P]),
    "r3g_tex_interpolate": (_I, [_P, _P, _I, _P, _P, _P, ctypes.c_int64, _P, _P]),
    "r3g_tex_view_weight": (_I, [_P, _P, _P, _P, _I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P]),
    "r3g_tex_bake": (_I, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_int64, _I, _P, _P]),
    "r3g_tex_bake_gather": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, ctypes.c_float, _P, _P]),
    "r3g_tex_bake_finalize": (_I, [_P, _P, _I, _P, _P, _P]),
    "r3g_tex_inpaint": (_I, [_P, _P, _P, _I, _P, _P, _P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _I, ctypes.POINTER(_I), _P]),
    "r3g_model_create": (_I, [_P, _P]),
    "r3g_model_set_tensor": (_I, [_P, ctypes.c_char_p, _P, _I, ctypes.c_int64, ctypes.c_int64]),
    "r3g_model_set_scalar": (_I, [_P, ctypes.c_char_p, ctypes.c_float]),
    "r3g_model_trim": (_I, [_P]),
    "r3g_cond_encode": (_I, [_P, _P, _P, _P]),
    "r3g_dit_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "r3g_dit_stream": (_I, [_P, _P, _I, _P]),
    "r3g_flow_sample": (_I, [_P, _P, _P, _I, ctypes.c_float, ctypes.c_float, _I, _P]),
    "r3g_flow_sample_batch": (_I, [_P, _P, _P, _I, _I, ctypes.c_float, ctypes.c_float, _I, _P]),
    "r3g_vae_decode": (_I, [_P, _P, _P, _P]),
    "r3g_grid_query": (_I, [_P, _D, _I, _P, ctypes.c_int64, ctypes.c_int64, _P]),
    "r3g_unet_create": (_I, [_P, _P]),
    "r3g_unet_set_tensor": (_I, [_P, ctypes.c_char_p, _P, _I, ctypes.c_int64, ctypes.c_int64]),
    "r3g_unet_resnet": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _I, _P, _P, _P]),
    "r3g_unet_transformer": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _P, _I, _P]),
    "r3g_unet_downsample": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _P, _P]),
    "r3g_unet_down_block": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "r3g_unet_forward": (_I, [_P, _P, _I, _I, ctypes.c_float, _P, _I, _P, _P]),
    "r3g_unet_mid_block": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "r3g_unet_forward_mv": (_I, [_P, _P, _I, _I, ctypes.c_float, _P, _I, _I, _P, _I, ctypes.c_float, ctypes.c_float, _P, _P]),
    "r3g_unet_transformer_mv": (_I, [_P, ctypes.c_char_p, _P, _I, _I, _I, _P, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P]),
    "r3g_unet_condition": (_I, [_P, ctypes.c_char_p, _P, _P, _P]),
    "r3g_aekl_decode": (_I, [_P, _P, _I, _I, _P, _P]),
    "r3g_aekl_encode": (_I, [_P, _P, _I, _I, _P, _P]),
    "r3g_sched_pix2pix_input": (_I, [_P, _P, _I, ctypes.c_int64, ctypes.c_float, _P, _P]),
    "r3g_sched_model_input": (_I, [_P, _I, _P, _I, ctypes.c_int64, ctypes.c_float, _P, _P]),
    "r3g_sched_cfg_combine": (_I, [_P, _P, ctypes.c_int64, ctypes.c_float, _P, _P]),
    "r3g_sched_euler_ancestral_step": (_I, [_P, _P, _P, ctypes.c_int64, ctypes.c_float, ctypes.c_float, _I, _P]),
    "r3g_op_gemm": (_I, [_P, ctypes.c_int64, _P, ctypes.c_int64, _P, _P, ctypes.c_int64, _P, _I, _I, _I, _I, _I, _P]),
    "r3g_op_gemm_splitk": (_I, [_P, ctypes.c_int64, _P, ctypes.c_int64, _P, _P, ctypes.c_int64, _P, _I, _I, _I, _I, _P, ctypes.c_int64, _P, _P]),
    "r3g_op_quant_fp8": (_I, [_P, ctypes.c_int64, _I, _I, _P, ctypes.c_int64, _P, _P]),
    "r3g_op_gemm_fp8": (_I, [_P, ctypes.c_int64, _P, _P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P, _I, _I, _I, _I, _P]),
    "r3g_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "r3g_set_staging": (_I, [_I]),
    "r3g_set_option": (_I, [ctypes.c_char_p, _I]),
    "r3g_get_counter": (_I, [ctypes.c_char_p, _P]),
    "r3g_prof_enable": (_I, [_I]),
    "r3g_prof_read": (_I, [_P, _P, _P, _I]),
    "r3g_prof_read_bytes": (_I, [_P, _I]),
}


class ModelConfig(ctypes.Structure):
    """struct r3g_model_config (include/r3g.h)"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "dit_in_channels", "dit_context_dim", "dit_hidden", "dit_heads", "dit_depth_double", "dit_depth_single",
        "dit_mlp_hidden", "dit_qkv_bias")] + [("dit_time_factor", ctypes.c_float)] + [(n, ctypes.c_int32) for n in (
            "vae_num_latents", "vae_embed_dim", "vae_width", "vae_heads", "vae_layers", "vae_num_freqs",
            "vae_include_pi", "vae_qkv_bias", "vae_qk_norm", "vae_mlp_ratio", "vae_ln_post")] + [
        ("vae_scale_factor", ctypes.c_float)] + [(n, ctypes.c_int32) for n in (
            "cond_image_size", "cond_patch", "cond_hidden", "cond_layers", "cond_heads", "cond_ffn_hidden")] + [
        ("cond_ln_eps", ctypes.c_float), ("grid_chunk", ctypes.c_int32)]

class UnetConfig(ctypes.Structure):
    """struct r3g_unet_config (include/r3g.h)"""
    _fields_ = [(n, ctypes.c_int32) for n in ("max_hw", "max_channels", "temb_dim", "ctx_dim", "ctx_tokens", "groups")] + [
        ("resnet_eps", ctypes.c_float)] + [(n, ctypes.c_int32) for n in ("n_levels", "layers_per_block", "in_channels", "out_channels")] + [
        ("block_out_channels", ctypes.c_int32 * 4)]


_LIB = None
_CTX = {}


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libr3g.so not built: run `python 3d-re-gen_amd/build.py` "
                              "(or __graft_entry__.build()); there is no fallback path")
        # PyTorch ships its own HIP runtime: it has to be in the process before libr3g.so is, so that both resolve
        # to the same libamdhip64 (two runtimes in one process: the second one finds "no ROCm-capable device")
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
        # ablation switches, e.g. R3G_OPTIONS="gemm_waves=8,fuse_qkv=0"
        for item in filter(None, os.environ.get("R3G_OPTIONS", "").split(",")):
            k, v = item.split("=")
            if L.r3g_set_option(k.strip().encode(), int(v)) != 0:
                raise ValueError("R3G_OPTIONS: " + L.r3g_last_error().decode())
    return _LIB


def check(rc):
    if rc == 0:
        return
    msg = lib().r3g_last_error().decode("utf-8", "replace")
    if rc == R3G_ERR_LEVEL_RANGE:
        raise LevelRangeError(msg)
    if rc == R3G_ERR_NO_SURFACE:
        raise NoSurfaceError(msg)
    raise R3GError(rc, msg)


def counter(name):
    """r3g_get_counter: a process-wide event counter of the library ("dit_f16_fallbacks", "dit_groups")"""
    v = ctypes.c_int64(0)
    check(lib().r3g_get_counter(name.encode(), ctypes.byref(v)))
    return int(v.value)


def new_context(device=0):
    """A further r3g_ctx on a device: its own model slot and workspaces, for a second pipeline that runs concurrently with
    the first (one context is not thread-safe; two are independent).  The caller keeps the handle."""
    lib()
    h = _P()
    check(lib().r3g_create(int(device), ctypes.byref(h)))
    return h


_LOCKS = {}
_LOCKS_GUARD = __import__("threading").Lock()


def device_lock(device=0):
    """One re-entrant lock per device for calls that go through the SHARED context: its mesh / texture workspaces and its
    pinned read-back buffer belong to one call at a time.  Pipelines with private contexts run their shape model
    concurrently, but their threads still meet in the mesh cleaners and the texture stage, which use the shared context."""
    with _LOCKS_GUARD:
        if device not in _LOCKS:
            _LOCKS[device] = __import__("threading").RLock()
        return _LOCKS[device]


def context(device=0):
    """The process-wide r3g_ctx of a device (created on first use)."""
    if device not in _CTX:
        h = _P()
        check(lib().r3g_create(int(device), ctypes.byref(h)))
        _CTX[device] = h
    return _CTX[device]

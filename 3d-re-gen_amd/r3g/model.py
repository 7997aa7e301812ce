"""Host side of the shape model: configuration, checkpoint -> device weights, and thin methods over
the C ABI (include/r3g.h).  Mirrors what `Hunyuan3DDiTFlowMatchingPipeline.from_pretrained` builds
from config.yaml + model.safetensors (reference src/2d_to_3d_models/run.py:122-124, 204-206).

The state dict uses upstream key names with the 'model.' / 'vae.' / 'conditioner.' prefixes.
"""
import ctypes

import torch

from . import ffi as _l


def swiglu_hidden(hidden_size, mlp_ratio):
    """transformers Dinov2SwiGLUFFN: hidden_features = (int(hidden*mlp_ratio*2/3) + 7) // 8 * 8"""
    return (int(int(hidden_size * mlp_ratio) * 2 / 3) + 7) // 8 * 8


def make_config(cfg, grid_chunk=0):
    d, v, c = cfg["dit"], cfg["vae"], cfg["cond"]
    if not c.get("use_swiglu_ffn", True):
        raise ValueError("only the SwiGLU Dinov2 variant (dinov2-giant) is implemented")
    if d.get("guidance_embed", False):
        raise ValueError("guidance-distilled DiT variants are not on the reference's default path")
    m = _l.ModelConfig()
    m.dit_in_channels, m.dit_context_dim, m.dit_hidden = d["in_channels"], d["context_in_dim"], d["hidden_size"]
    m.dit_heads, m.dit_depth_double, m.dit_depth_single = d["num_heads"], d["depth"], d["depth_single_blocks"]
    m.dit_mlp_hidden = int(d["hidden_size"] * d["mlp_ratio"])
    m.dit_qkv_bias, m.dit_time_factor = int(d["qkv_bias"]), float(d["time_factor"])
    m.vae_num_latents, m.vae_embed_dim, m.vae_width, m.vae_heads = v["num_latents"], v["embed_dim"], v["width"], v["heads"]
    m.vae_layers, m.vae_num_freqs, m.vae_include_pi = v["num_decoder_layers"], v["num_freqs"], int(v["include_pi"])
    m.vae_qkv_bias, m.vae_qk_norm = int(v["qkv_bias"]), int(v["qk_norm"])
    m.vae_mlp_ratio, m.vae_ln_post = int(v.get("geo_decoder_mlp_expand_ratio", 4)), int(v.get("geo_decoder_ln_post", True))
    m.vae_scale_factor = float(v["scale_factor"])
    m.cond_image_size, m.cond_patch, m.cond_hidden = c["image_size"], c["patch_size"], c["hidden_size"]
    m.cond_layers, m.cond_heads = c["num_hidden_layers"], c["num_attention_heads"]
    m.cond_ffn_hidden = swiglu_hidden(c["hidden_size"], c["mlp_ratio"])
    m.cond_ln_eps = float(c.get("layer_norm_eps", 1e-6))
    m.grid_chunk = int(grid_chunk)
    return m


def _pad_k(w):
    n, k = w.shape
    kp = (k + 63) // 64 * 64
    if kp == k:
        return w
    out = torch.zeros((n, kp), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


def prepare_weights(sd, device):
    """upstream state dict -> {name: (tensor on device, dtype code)}; matrices bf16 [N][Kpad64], vectors f32.
    Fusions done here (pure re-layout, no arithmetic): Dinov2 query/key/value -> one qkv matrix; patch conv
    kernel [C,3,p,p] -> [C, 3*p*p]; cls/pos embeddings flattened."""
    out, scalars = {}, {}
    qkv_parts = {}
    for k, t in sd.items():
        if not torch.is_floating_point(t) or k.endswith("mask_token"):
            continue
        t = t.detach()
        if ".attention.attention." in k and k.split(".")[-2] in ("query", "key", "value"):
            base, which, kind = k.rsplit(".", 2)
            qkv_parts.setdefault((base, kind), {})[which] = t
            continue
        if k.endswith("patch_embeddings.projection.weight"):
            t = t.reshape(t.shape[0], -1)
        if k.endswith("cls_token"):
            t = t.reshape(-1)
        if k.endswith("position_embeddings"):
            t = t.reshape(t.shape[-2], t.shape[-1])
        if k.endswith("geo_decoder.output_proj.bias"):
            scalars[k] = float(t.reshape(-1)[0])
            continue
        if k.endswith("geo_decoder.output_proj.weight"):
            out[k] = (t.reshape(1, -1).to(device=device, dtype=torch.float32).contiguous(), 0)
            continue
        if t.ndim == 2 and not k.endswith("position_embeddings"):
            out[k] = (_pad_k(t.to(device=device, dtype=torch.float32)).to(torch.bfloat16).contiguous(), 1)
        else:
            out[k] = (t.to(device=device, dtype=torch.float32).reshape(1, -1).contiguous(), 0)
    for (base, kind), parts in qkv_parts.items():
        t = torch.cat([parts["query"], parts["key"], parts["value"]], dim=0)
        name = base + ".qkv." + kind
        if kind == "weight":
            out[name] = (_pad_k(t.to(device=device, dtype=torch.float32)).to(torch.bfloat16).contiguous(), 1)
        else:
            out[name] = (t.to(device=device, dtype=torch.float32).reshape(1, -1).contiguous(), 0)
    return out, scalars


class ShapeModel:
    """Weights + activation arena of one shape model on one GPU."""

    def __init__(self, cfg, state_dict, device=0, grid_chunk=0, private_ctx=False):
        if not torch.cuda.is_available():
            raise RuntimeError("r3g.ShapeModel needs an MI355X: libr3g has no CPU path")
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.private_ctx = bool(private_ctx)
        self.ctx = _l.new_context(device) if private_ctx else _l.context(device)
        self.L = _l.lib()
        self._c = make_config(cfg, grid_chunk)
        with torch.cuda.device(self.device):
            self._w, self._scalars = prepare_weights(state_dict, self.device)
            self._install()
        p = cfg["cond"]["image_size"] // cfg["cond"]["patch_size"]
        self.cond_tokens = p * p + 1
        self.num_latents = cfg["vae"]["num_latents"]
        self.in_channels = cfg["dit"]["in_channels"]

    # The C context holds ONE model (arena + weight table) per device.  Several ShapeModel objects may coexist in a
    # process (e.g. mini + full, or test fixtures): the one being called re-installs itself if it is not current.
    _current = {}

    def _install(self):
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _l.check(self.L.r3g_model_create(self.ctx, ctypes.byref(self._c)))
            for name, (t, code) in self._w.items():
                _l.check(self.L.r3g_model_set_tensor(self.ctx, name.encode(), t.data_ptr(), code, t.shape[0], t.shape[1]))
            for name, v in self._scalars.items():
                _l.check(self.L.r3g_model_set_scalar(self.ctx, name.encode(), v))
            torch.cuda.synchronize()
        if not self.private_ctx:
            ShapeModel._current[self.device.index] = self
        self._have_z = False

    def trim(self):
        """give the query-side cache of the geo decoder back to the device (r3g_model_trim); it is rebuilt on the next grid query"""
        if self.private_ctx or ShapeModel._current.get(self.device.index) is self:
            with torch.cuda.device(self.device):
                _l.check(self.L.r3g_model_trim(self.ctx))

    def _activate(self):
        if not self.private_ctx and ShapeModel._current.get(self.device.index) is not self:
            self._install()

    def _s(self):
        self._activate()
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def cond_encode(self, image):
        """image f32 [3,S,S] (resized / cropped / ImageNet-normalised) -> bf16 [tokens, hidden]"""
        image = image.to(self.device, torch.float32).contiguous()
        out = torch.empty((self.cond_tokens, self.cfg["cond"]["hidden_size"]), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_cond_encode(self.ctx, image.data_ptr(), out.data_ptr(), self._s()))
        return out

    def dit_forward(self, x, t, cond, n_double=-1, n_single=-1):
        """x f32 [B,N,C], t f32 [B], cond bf16 [B,Lc,D] -> f32 [B,N,C]"""
        x = x.to(self.device, torch.float32).contiguous()
        t = t.to(self.device, torch.float32).contiguous()
        cond = cond.to(self.device, torch.bfloat16).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_dit_forward(self.ctx, x.data_ptr(), t.data_ptr(), cond.data_ptr(), out.data_ptr(),
                                            x.shape[0], n_double, n_single, self._s()))
        return out

    def dit_stream(self, batch):
        """joint residual stream left by the last dit_forward(): f32 [B, cond_tokens + num_latents, hidden], cond first"""
        out = torch.empty((batch, self.cond_tokens + self.num_latents, self.cfg["dit"]["hidden_size"]),
                          dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_dit_stream(self.ctx, out.data_ptr(), int(batch), self._s()))
        return out

    def flow_sample(self, latents, cond2, steps, guidance_scale, shift=1.0, uncond_uniform=None):
        """latents f32 [N,C] (modified in place and returned), cond2 bf16 [2,Lc,D] = [cond, uncond].
        uncond_uniform: all unconditional tokens identical (None = check on the device)."""
        latents = latents.to(self.device, torch.float32).contiguous()
        cond2 = cond2.to(self.device, torch.bfloat16).contiguous()
        if uncond_uniform is None:
            uncond_uniform = bool((cond2[1] == cond2[1, :1]).all().item())
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_flow_sample(self.ctx, latents.data_ptr(), cond2.data_ptr(), int(steps),
                                            float(guidance_scale), float(shift), int(bool(uncond_uniform)), self._s()))
        return latents

    def flow_sample_batch(self, latents, cond2, steps, guidance_scale, shift=1.0, uncond_uniform=None):
        """n independent objects through the denoising loop together: latents f32 [n,N,C] (modified in place and
        returned), cond2 bf16 [n,2,Lc,D].  Per-object results equal flow_sample() of that object bit for bit."""
        latents = latents.to(self.device, torch.float32).contiguous()
        cond2 = cond2.to(self.device, torch.bfloat16).contiguous()
        if latents.ndim != 3 or cond2.ndim != 4 or cond2.shape[0] != latents.shape[0] or cond2.shape[1] != 2:
            raise ValueError("flow_sample_batch: latents [n,N,C] and cond2 [n,2,Lc,D] expected")
        if uncond_uniform is None:
            uncond_uniform = bool((cond2[:, 1] == cond2[:, 1, :1]).all().item())
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_flow_sample_batch(self.ctx, latents.data_ptr(), cond2.data_ptr(), int(latents.shape[0]),
                                                  int(steps), float(guidance_scale), float(shift),
                                                  int(bool(uncond_uniform)), self._s()))
        return latents

    def vae_decode(self, latents, return_z=False):
        latents = latents.to(self.device, torch.float32).contiguous()
        z = torch.empty((self.num_latents, self.cfg["vae"]["width"]), dtype=torch.float32, device=self.device) \
            if return_z else None
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_vae_decode(self.ctx, latents.data_ptr(), z.data_ptr() if return_z else None, self._s()))
        self._have_z = True
        return z

    def grid_query(self, bound, octree_resolution, out=None, start=0, count=None):
        n = octree_resolution + 1
        if out is None:
            out = torch.empty((n, n, n), dtype=torch.float32, device=self.device)
        if count is None:
            count = n ** 3 - start
        with torch.cuda.device(self.device):
            _l.check(self.L.r3g_grid_query(self.ctx, float(bound), int(octree_resolution), out.data_ptr(), int(start),
                                           int(count), self._s()))
        return out

"""oracle/hy3d_torch.py -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).

PyTorch-CPU fp32 restatement of the Hunyuan3D-2 shape pipeline that the reference stage drives:
    reference call sites: src/2d_to_3d_models/run.py:10-17 (imports), :77-84 (pipeline call),
                          :122-124 / :204-206 (from_pretrained)
    arithmetic lives in : the un-vendored git submodule Tencent/Hunyuan3D-2 (`hy3dgen`),
                          pin unknown (.gitmodules:4-6; the directory is empty in this snapshot).

Because neither the upstream source nor any weights are present in the build container, this file
restates the PUBLISHED upstream modules from recollection of their structure
(hy3dgen/shapegen/{pipelines,schedulers,preprocessors}.py, models/denoisers/hunyuan3ddit.py,
models/autoencoders/{model,attention_blocks,volume_decoders}.py, models/conditioner.py), keeping
upstream module / parameter names so a real `model.fp16.safetensors` would load with strict=True.
"Parity unpinned" = no golden vector from the reference exists for the DiT / VAE / grid query;
the HIP kernels are checked against THIS file on seeded synthetic weights.  Only the
marching-cubes tail (oracle/mc_lewiner.c) is pinned to the real dependency.
The conditioner uses the real `transformers.Dinov2Model` (the class upstream instantiates).

Every dimension comes from a config dict shaped like upstream's config.yaml (`full_config()`,
`mini_config()`, `tiny_config()`).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- configs
def full_config():
    """hunyuan3d-dit-v2-0 (recalled; reproduces the published 1.1 B DiT parameter count)."""
    return dict(
        dit=dict(in_channels=64, context_in_dim=1536, hidden_size=1024, mlp_ratio=4.0, num_heads=16, depth=16,
                 depth_single_blocks=32, qkv_bias=True, time_factor=1000, guidance_embed=False),
        vae=dict(num_latents=3072, embed_dim=64, width=1024, heads=16, num_decoder_layers=16, num_freqs=8,
                 include_pi=False, qkv_bias=False, qk_norm=True, scale_factor=0.9990943042622529,
                 geo_decoder_mlp_expand_ratio=4, geo_decoder_ln_post=True),
        cond=dict(image_size=518, patch_size=14, hidden_size=1536, num_hidden_layers=40, num_attention_heads=24,
                  mlp_ratio=4, use_swiglu_ffn=True, layer_norm_eps=1e-6, layerscale_value=1.0),
        sched=dict(num_train_timesteps=1000, shift=1.0),
        proc=dict(size=512, border_ratio=0.15),
        guidance_scale=5.0, box_v=1.01, mc_level=0.0)


def wide_config(depth=1, depth_single=1, vae_layers=1, cond_layers=1):
    """Full widths / token counts of hunyuan3d-dit-v2-0 with reduced depth (for full-shape parity tests)."""
    c = full_config()
    c["dit"].update(depth=depth, depth_single_blocks=depth_single)
    c["vae"].update(num_decoder_layers=vae_layers)
    c["cond"].update(num_hidden_layers=cond_layers)
    return c


def mini_config():
    """hunyuan3d-dit-v2-mini (recalled, lower confidence): half the depth, 512 latents."""
    c = full_config()
    c["dit"].update(depth=8, depth_single_blocks=16)
    c["vae"].update(num_latents=512)
    return c


def tiny_config():
    """CI-sized: same structure, every dimension small (SURVEY.md 8d 'tiny-dim').  Head dim stays 64
    everywhere and the Dinov2 SwiGLU width ((int(192*4*2/3)+7)//8*8 = 512) stays a multiple of 64."""
    return dict(
        dit=dict(in_channels=16, context_in_dim=192, hidden_size=128, mlp_ratio=4.0, num_heads=2, depth=2,
                 depth_single_blocks=3, qkv_bias=True, time_factor=1000, guidance_embed=False),
        vae=dict(num_latents=256, embed_dim=16, width=128, heads=2, num_decoder_layers=2, num_freqs=8,
                 include_pi=False, qkv_bias=False, qk_norm=True, scale_factor=0.9990943042622529,
                 geo_decoder_mlp_expand_ratio=4, geo_decoder_ln_post=True),
        cond=dict(image_size=70, patch_size=14, hidden_size=192, num_hidden_layers=2, num_attention_heads=3,
                  mlp_ratio=4, use_swiglu_ffn=True, layer_norm_eps=1e-6, layerscale_value=1.0),
        sched=dict(num_train_timesteps=1000, shift=1.0),
        proc=dict(size=64, border_ratio=0.15),
        guidance_scale=5.0, box_v=1.01, mc_level=0.0)


# ----------------------------------------------------------------------------- DiT (hunyuan3ddit.py)
def timestep_embedding(t, dim, max_period=10000, time_factor=1000.0):
    t = time_factor * t
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class MLPEmbedder(nn.Module):
    def __init__(self, in_dim, hidden_dim):
        super().__init__()
        self.in_layer = nn.Linear(in_dim, hidden_dim, bias=True)
        self.silu = nn.SiLU()
        self.out_layer = nn.Linear(hidden_dim, hidden_dim, bias=True)

    def forward(self, x):
        return self.out_layer(self.silu(self.in_layer(x)))


class RMSNorm(nn.Module):
    eps = 1e-6

    def __init__(self, dim):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        rrms = torch.rsqrt(torch.mean(x.float() ** 2, dim=-1, keepdim=True) + self.eps)
        return (x.float() * rrms).to(x.dtype) * self.scale


class QKNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.query_norm = RMSNorm(dim)
        self.key_norm = RMSNorm(dim)

    def forward(self, q, k, v):
        return self.query_norm(q).to(v), self.key_norm(k).to(v)


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.norm = QKNorm(dim // num_heads)
        self.proj = nn.Linear(dim, dim)


class Modulation(nn.Module):
    def __init__(self, dim, double):
        super().__init__()
        self.is_double = double
        self.multiplier = 6 if double else 3
        self.lin = nn.Linear(dim, self.multiplier * dim, bias=True)

    def forward(self, vec):
        out = self.lin(F.silu(vec))[:, None, :].chunk(self.multiplier, dim=-1)
        return out[:3], (out[3:] if self.is_double else None)


def _split_khd(qkv, heads):
    """ "B L (K H D) -> K B H L D", K=3 """
    B, L, _ = qkv.shape
    return qkv.view(B, L, 3, heads, -1).permute(2, 0, 3, 1, 4)


def _sdpa(q, k, v):
    return F.scaled_dot_product_attention(q, k, v)


def _attention(q, k, v):
    """attention(): SDPA then "B H L D -> B L (H D)" (no positional term on this path: pe=None)"""
    x = _sdpa(q, k, v)
    B, H, L, D = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * D)


def _modulate(x, shift, scale):
    return (1 + scale) * x + shift


def _joint(txt, img):
    """joint sequence of a double block / the single blocks: conditioning tokens FIRST"""
    return torch.cat((txt, img), dim=-2)


def _unjoint(x, n_txt):
    return x[..., :n_txt, :], x[..., n_txt:, :]


def _mlp(hidden, mlp_hidden):
    return nn.Sequential(nn.Linear(hidden, mlp_hidden, bias=True), nn.GELU(approximate="tanh"),
                         nn.Linear(mlp_hidden, hidden, bias=True))


class DoubleStreamBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio, qkv_bias):
        super().__init__()
        mlp_hidden = int(hidden_size * mlp_ratio)
        self.num_heads = num_heads
        self.img_mod = Modulation(hidden_size, True)
        self.img_norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.img_attn = SelfAttention(hidden_size, num_heads, qkv_bias)
        self.img_norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.img_mlp = _mlp(hidden_size, mlp_hidden)
        self.txt_mod = Modulation(hidden_size, True)
        self.txt_norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.txt_attn = SelfAttention(hidden_size, num_heads, qkv_bias)
        self.txt_norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.txt_mlp = _mlp(hidden_size, mlp_hidden)

    def forward(self, img, txt, vec):
        (i_sh1, i_sc1, i_g1), (i_sh2, i_sc2, i_g2) = self.img_mod(vec)
        (t_sh1, t_sc1, t_g1), (t_sh2, t_sc2, t_g2) = self.txt_mod(vec)
        img_q, img_k, img_v = _split_khd(self.img_attn.qkv(_modulate(self.img_norm1(img), i_sh1, i_sc1)), self.num_heads)
        img_q, img_k = self.img_attn.norm(img_q, img_k, img_v)
        txt_q, txt_k, txt_v = _split_khd(self.txt_attn.qkv(_modulate(self.txt_norm1(txt), t_sh1, t_sc1)), self.num_heads)
        txt_q, txt_k = self.txt_attn.norm(txt_q, txt_k, txt_v)
        attn = _attention(_joint(txt_q, img_q), _joint(txt_k, img_k), _joint(txt_v, img_v))
        txt_attn, img_attn = _unjoint(attn, txt.shape[1])
        img = img + i_g1 * self.img_attn.proj(img_attn)
        img = img + i_g2 * self.img_mlp(_modulate(self.img_norm2(img), i_sh2, i_sc2))
        txt = txt + t_g1 * self.txt_attn.proj(txt_attn)
        txt = txt + t_g2 * self.txt_mlp(_modulate(self.txt_norm2(txt), t_sh2, t_sc2))
        return img, txt


class SingleStreamBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.mlp_hidden_dim = int(hidden_size * mlp_ratio)
        self.linear1 = nn.Linear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim)
        self.linear2 = nn.Linear(hidden_size + self.mlp_hidden_dim, hidden_size)
        self.norm = QKNorm(hidden_size // num_heads)
        self.pre_norm = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp_act = nn.GELU(approximate="tanh")
        self.modulation = Modulation(hidden_size, False)

    def forward(self, x, vec):
        (shift, scale, gate), _ = self.modulation(vec)
        x_mod = _modulate(self.pre_norm(x), shift, scale)
        qkv, mlp = torch.split(self.linear1(x_mod), [3 * self.hidden_size, self.mlp_hidden_dim], dim=-1)
        q, k, v = _split_khd(qkv, self.num_heads)
        q, k = self.norm(q, k, v)
        attn = _attention(q, k, v)
        return x + gate * self.linear2(torch.cat((attn, self.mlp_act(mlp)), 2))


class LastLayer(nn.Module):
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, vec):
        shift, scale = self.adaLN_modulation(vec).chunk(2, dim=1)
        return self.linear(_modulate(self.norm_final(x), shift[:, None, :], scale[:, None, :]))


class Hunyuan3DDiT(nn.Module):
    def __init__(self, in_channels, context_in_dim, hidden_size, mlp_ratio, num_heads, depth, depth_single_blocks,
                 qkv_bias, time_factor, guidance_embed=False):
        super().__init__()
        assert not guidance_embed, "distilled (guidance_embed) variants are not on the reference's default path"
        self.in_channels = in_channels
        self.time_factor = time_factor
        self.latent_in = nn.Linear(in_channels, hidden_size, bias=True)
        self.time_in = MLPEmbedder(256, hidden_size)
        self.cond_in = nn.Linear(context_in_dim, hidden_size)
        self.double_blocks = nn.ModuleList(
            [DoubleStreamBlock(hidden_size, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.single_blocks = nn.ModuleList(
            [SingleStreamBlock(hidden_size, num_heads, mlp_ratio) for _ in range(depth_single_blocks)])
        self.final_layer = LastLayer(hidden_size, 1, in_channels)

    def forward(self, x, t, cond, n_double=None, n_single=None):
        latent = self.latent_in(x)
        vec = self.time_in(timestep_embedding(t, 256, time_factor=self.time_factor).to(latent.dtype))
        cond = self.cond_in(cond)
        for blk in self.double_blocks[:n_double]:
            latent, cond = blk(latent, cond, vec)
        latent = _joint(cond, latent)
        for blk in self.single_blocks[:n_single]:
            latent = blk(latent, vec)
        latent = _unjoint(latent, cond.shape[1])[1]
        return self.final_layer(latent, vec)


# ----------------------------------------------------------------------------- ShapeVAE (autoencoders/)
class FourierEmbedder(nn.Module):
    def __init__(self, num_freqs=6, input_dim=3, include_input=True, include_pi=True):
        super().__init__()
        freqs = 2.0 ** torch.arange(num_freqs, dtype=torch.float32)
        if include_pi:
            freqs = freqs * torch.pi
        self.register_buffer("frequencies", freqs, persistent=False)
        self.include_input = include_input
        self.num_freqs = num_freqs
        self.out_dim = input_dim * (num_freqs * 2 + (1 if include_input or num_freqs == 0 else 0))

    def forward(self, x):
        embed = (x[..., None].contiguous() * self.frequencies).view(*x.shape[:-1], -1)
        if self.include_input:
            return torch.cat((x, embed.sin(), embed.cos()), dim=-1)
        return torch.cat((embed.sin(), embed.cos()), dim=-1)


class MLP(nn.Module):
    def __init__(self, width, expand_ratio=4):
        super().__init__()
        self.c_fc = nn.Linear(width, width * expand_ratio)
        self.c_proj = nn.Linear(width * expand_ratio, width)
        self.gelu = nn.GELU()

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class QKVMultiheadCrossAttention(nn.Module):
    def __init__(self, heads, width, qk_norm):
        super().__init__()
        self.heads = heads
        self.q_norm = nn.LayerNorm(width // heads, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
        self.k_norm = nn.LayerNorm(width // heads, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()

    def forward(self, q, kv):
        _, n_ctx, _ = q.shape
        bs, n_data, width = kv.shape
        attn_ch = width // self.heads // 2
        q = q.view(bs, n_ctx, self.heads, -1)
        kv = kv.view(bs, n_data, self.heads, -1)
        k, v = torch.split(kv, attn_ch, dim=-1)
        q, k = self.q_norm(q), self.k_norm(k)
        q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(bs, n_ctx, -1)


class MultiheadCrossAttention(nn.Module):
    def __init__(self, width, heads, qkv_bias, qk_norm):
        super().__init__()
        self.c_q = nn.Linear(width, width, bias=qkv_bias)
        self.c_kv = nn.Linear(width, width * 2, bias=qkv_bias)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadCrossAttention(heads, width, qk_norm)

    def forward(self, x, data):
        return self.c_proj(self.attention(self.c_q(x), self.c_kv(data)))


class ResidualCrossAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_expand_ratio, qkv_bias, qk_norm):
        super().__init__()
        self.attn = MultiheadCrossAttention(width, heads, qkv_bias, qk_norm)
        self.ln_1 = nn.LayerNorm(width, elementwise_affine=True, eps=1e-6)
        self.ln_2 = nn.LayerNorm(width, elementwise_affine=True, eps=1e-6)
        self.ln_3 = nn.LayerNorm(width, elementwise_affine=True, eps=1e-6)
        self.mlp = MLP(width, mlp_expand_ratio)

    def forward(self, x, data):
        x = x + self.attn(self.ln_1(x), self.ln_2(data))
        return x + self.mlp(self.ln_3(x))


class QKVMultiheadAttention(nn.Module):
    def __init__(self, heads, width, qk_norm):
        super().__init__()
        self.heads = heads
        self.q_norm = nn.LayerNorm(width // heads, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()
        self.k_norm = nn.LayerNorm(width // heads, elementwise_affine=True, eps=1e-6) if qk_norm else nn.Identity()

    def forward(self, qkv):
        bs, n_ctx, width = qkv.shape
        attn_ch = width // self.heads // 3
        qkv = qkv.view(bs, n_ctx, self.heads, -1)
        q, k, v = torch.split(qkv, attn_ch, dim=-1)   # per-head interleaved [h: (q,k,v)]
        q, k = self.q_norm(q), self.k_norm(k)
        q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(bs, n_ctx, -1)


class MultiheadAttention(nn.Module):
    def __init__(self, width, heads, qkv_bias, qk_norm):
        super().__init__()
        self.c_qkv = nn.Linear(width, width * 3, bias=qkv_bias)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadAttention(heads, width, qk_norm)

    def forward(self, x):
        return self.c_proj(self.attention(self.c_qkv(x)))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, qkv_bias, qk_norm):
        super().__init__()
        self.attn = MultiheadAttention(width, heads, qkv_bias, qk_norm)
        self.ln_1 = nn.LayerNorm(width, elementwise_affine=True, eps=1e-6)
        self.mlp = MLP(width)
        self.ln_2 = nn.LayerNorm(width, elementwise_affine=True, eps=1e-6)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, qkv_bias, qk_norm):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, qkv_bias, qk_norm) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


class CrossAttentionDecoder(nn.Module):
    def __init__(self, fourier_embedder, width, heads, mlp_expand_ratio, enable_ln_post, qkv_bias, qk_norm):
        super().__init__()
        self.fourier_embedder = fourier_embedder
        self.enable_ln_post = enable_ln_post
        self.query_proj = nn.Linear(fourier_embedder.out_dim, width)
        if not enable_ln_post:
            qk_norm = False
        self.cross_attn_decoder = ResidualCrossAttentionBlock(width, heads, mlp_expand_ratio, qkv_bias, qk_norm)
        if enable_ln_post:
            self.ln_post = nn.LayerNorm(width)
        self.output_proj = nn.Linear(width, 1)

    def forward(self, queries, latents):
        x = self.cross_attn_decoder(self.query_proj(self.fourier_embedder(queries).to(latents.dtype)), latents)
        if self.enable_ln_post:
            x = self.ln_post(x)
        return self.output_proj(x)


class ShapeVAE(nn.Module):
    def __init__(self, num_latents, embed_dim, width, heads, num_decoder_layers, num_freqs, include_pi, qkv_bias,
                 qk_norm, scale_factor, geo_decoder_mlp_expand_ratio=4, geo_decoder_ln_post=True):
        super().__init__()
        self.latent_shape = (num_latents, embed_dim)
        self.scale_factor = scale_factor
        self.fourier_embedder = FourierEmbedder(num_freqs=num_freqs, include_pi=include_pi)
        self.post_kl = nn.Linear(embed_dim, width)
        self.transformer = Transformer(width, num_decoder_layers, heads, qkv_bias, qk_norm)
        self.geo_decoder = CrossAttentionDecoder(self.fourier_embedder, width, heads, geo_decoder_mlp_expand_ratio,
                                                 geo_decoder_ln_post, qkv_bias, qk_norm)

    def forward(self, latents):
        return self.transformer(self.post_kl(latents))


def dense_grid_points(bound, octree_resolution):
    """volume_decoders.generate_dense_grid_points, indexing='ij': point index = (i*(R+1)+j)*(R+1)+k."""
    x = np.linspace(-bound, bound, int(octree_resolution) + 1, dtype=np.float32)
    xs, ys, zs = np.meshgrid(x, x, x, indexing="ij")
    return np.stack((xs, ys, zs), axis=-1).reshape(-1, 3)


@torch.no_grad()
def volume_decode(vae, latents, bound, octree_resolution, num_chunks):
    """VanillaVolumeDecoder.__call__ for batch 1 -> float32 grid [R+1]^3."""
    xyz = torch.from_numpy(dense_grid_points(bound, octree_resolution)).to(latents.dtype)
    out = []
    for s in range(0, xyz.shape[0], num_chunks):
        out.append(vae.geo_decoder(queries=xyz[None, s:s + num_chunks], latents=latents))
    n = int(octree_resolution) + 1
    return torch.cat(out, dim=1).view(n, n, n).float()


# ----------------------------------------------------------------------------- conditioner
class DinoImageEncoder(nn.Module):
    """conditioner.ImageEncoder/DinoImageEncoder: resize+centre-crop to image_size, ImageNet normalise,
    transformers.Dinov2Model, last_hidden_state incl. CLS.  Unconditional embedding = zeros."""
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]

    def __init__(self, cfg):
        super().__init__()
        from transformers import Dinov2Config, Dinov2Model
        kw = {k: v for k, v in cfg.items()}
        self.image_size = kw["image_size"]
        self.model = Dinov2Model(Dinov2Config(**kw))
        self.model.eval()
        self.num_patches = (self.image_size // cfg["patch_size"]) ** 2 + 1

    @staticmethod
    def transform(image, image_size, mean, std):
        """torchvision Resize(image_size, BILINEAR, antialias=True) + CenterCrop + Normalize on [B,3,H,W] in [0,1]."""
        _, _, h, w = image.shape
        s = image_size / min(h, w)
        nh, nw = (image_size, int(w * s)) if h <= w else (int(h * s), image_size)   # torchvision Resize truncates
        x = F.interpolate(image, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
        top, left = (nh - image_size) // 2, (nw - image_size) // 2
        x = x[:, :, top:top + image_size, left:left + image_size]
        m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
        sd = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
        return (x - m) / sd

    def forward(self, image, value_range=(-1, 1)):
        low, high = value_range
        image = (image - low) / (high - low)
        return self.model(self.transform(image, self.image_size, self.mean, self.std)).last_hidden_state


# ----------------------------------------------------------------------------- preprocess (preprocessors.py)
def _resize_u8(arr, w, h, resample):
    from PIL import Image
    if arr.ndim == 3 and arr.shape[2] == 1:
        return np.asarray(Image.fromarray(arr[..., 0]).resize((w, h), resample))[..., None]
    return np.asarray(Image.fromarray(arr).resize((w, h), resample))


def _area_axis(a, n_out):
    """shrink axis 0 of a float64 array by the pixel-area relation, written through the running integral of the signal: the mean
    over [i s, (i + 1) s) is (F((i + 1) s) - F(i s)) / s with F piecewise linear between the prefix sums (an independent
    formulation of what the product builds as gathered, weighted taps)"""
    n_in = a.shape[0]
    s = n_in / n_out
    F = np.concatenate([np.zeros((1,) + a.shape[1:]), np.cumsum(a, axis=0)], axis=0)           # F[k] = sum of the first k pixels

    def at(t):
        t = np.minimum(t, n_in)
        k = np.minimum(np.floor(t).astype(np.int64), n_in - 1)
        frac = (t - k).reshape((-1,) + (1,) * (a.ndim - 1))
        return F[k] + frac * a[k]
    lo = np.arange(n_out) * s
    hi = np.minimum(lo + s, n_in)
    return (at(hi) - at(lo)) / (hi - lo).reshape((-1,) + (1,) * (a.ndim - 1))


def _grow_axis(a, n_out):
    """OpenCV INTER_AREA along axis 0 when it grows ([RECALLED] imgproc/resize.cpp): two taps, coefficients fx = frac((d + 1) -
    (sx + 1) n_out / n_in) (0 when negative), as 11-bit integers; returns the integer sums at scale 2^11"""
    n_in = a.shape[0]
    out = np.empty((n_out,) + a.shape[1:], np.int64)
    for d in range(n_out):
        sx = int(np.floor(d * (n_in / n_out)))
        fx = np.float32((d + 1) - (sx + 1) * (n_out / n_in))
        fx = np.float32(0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx >= n_in - 1:
            sx, fx = n_in - 1, np.float32(0)
        c1 = int(np.rint(fx * np.float32(2048)))
        c0 = int(np.rint((np.float32(1) - fx) * np.float32(2048)))
        out[d] = a[sx] * c0 + a[min(sx + 1, n_in - 1)] * c1
    return out


def resize_area(arr, w, h):
    """cv2.INTER_AREA restated for uint8 [H][W][C] (cv2 is absent here; the product restates it independently as weighted taps)"""
    H, W = arr.shape[:2]
    if H == 2 * h and W == 2 * w:               # resizeAreaFast, 2 x 2: (a + b + c + d + 2) >> 2
        q = arr.astype(np.int64).reshape(h, 2, w, 2, -1).sum(axis=(1, 3))
        return ((q + 2) >> 2).astype(np.uint8)
    if w <= W and h <= H:
        out = _area_axis(arr.astype(np.float64), h)
        out = _area_axis(out.transpose(1, 0, 2), w).transpose(1, 0, 2)
        return np.rint(out).clip(0, 255).astype(np.uint8)
    a = arr.astype(np.int64)
    rows = _grow_axis(a.transpose(1, 0, 2), w).transpose(1, 0, 2)          # horizontal pass first, scale 2^11
    H0 = rows.shape[0]
    out = np.empty((h,) + rows.shape[1:], np.int64)
    for d in range(h):
        sy = int(np.floor(d * (H0 / h)))
        fy = np.float32((d + 1) - (sy + 1) * (h / H0))
        fy = np.float32(0) if fy <= 0 else np.float32(fy - np.floor(fy))
        if sy >= H0 - 1:
            sy, fy = H0 - 1, np.float32(0)
        b1 = int(np.rint(fy * np.float32(2048)))
        b0 = int(np.rint((np.float32(1) - fy) * np.float32(2048)))
        out[d] = (((rows[sy] >> 4) * b0) >> 16) + (((rows[min(sy + 1, H0 - 1)] >> 4) * b1) >> 16)
    return ((out + 2) >> 2).clip(0, 255).astype(np.uint8)


def recenter(image, border_ratio):
    """ImageProcessorV2.recenter; cv2.INTER_AREA restated (resize_area above; rounds 1-2 used PIL BOX)."""
    mask = image[..., 3]
    H, W, C = image.shape
    size = max(H, W)
    result = np.zeros((size, size, C), dtype=np.uint8)
    coords = np.nonzero(mask)
    x_min, x_max = coords[0].min(), coords[0].max()
    y_min, y_max = coords[1].min(), coords[1].max()
    h, w = x_max - x_min, y_max - y_min
    if h == 0 or w == 0:
        raise ValueError("input image is empty")
    desired = int(size * (1 - border_ratio))
    scale = desired / max(h, w)
    h2, w2 = int(h * scale), int(w * scale)
    x2 = (size - h2) // 2
    y2 = (size - w2) // 2
    result[x2:x2 + h2, y2:y2 + w2] = resize_area(image[x_min:x_max, y_min:y_max], w2, h2)
    bg = np.ones((size, size, 3), dtype=np.uint8) * 255
    m = result[..., 3:].astype(np.float32) / 255
    rgb = result[..., :3] * m + bg * (1 - m)
    return rgb.clip(0, 255).astype(np.uint8), (m * 255).clip(0, 255).astype(np.uint8)


def preprocess_image(pil_image, size, border_ratio):
    """ImageProcessorV2.__call__ -> image [1,3,size,size] in [-1,1], mask [1,1,size,size]."""
    from PIL import Image
    arr = np.asarray(pil_image.convert("RGBA"))
    rgb, mask = recenter(arr, border_ratio)
    rgb = _resize_u8(rgb, size, size, Image.BICUBIC)
    mask = _resize_u8(mask, size, size, Image.NEAREST)
    img = torch.tensor(np.ascontiguousarray(rgb)).float() / 255 * 2 - 1
    msk = torch.tensor(np.ascontiguousarray(mask)).float() / 255 * 2 - 1
    return img.permute(2, 0, 1)[None].contiguous(), msk.permute(2, 0, 1)[None].contiguous()


# ----------------------------------------------------------------------------- pipeline
def flow_sigmas(num_inference_steps, shift=1.0):
    """FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=linspace(0,1,N)): returns N+1 sigmas (last = 1)."""
    s = np.linspace(0, 1, num_inference_steps)
    s = shift * s / (1 + (shift - 1) * s)
    return np.concatenate([s.astype(np.float32), np.ones(1, np.float32)])


def prepare_latents(shape, generator, dtype=torch.float16):
    """diffusers randn_tensor with a CPU generator: drawn on the CPU in the PIPELINE dtype (upstream from_pretrained
    defaults to fp16; fp16 and fp32 draws consume the generator differently), then moved; computed on in fp32 here."""
    return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).float()


class ShapePipeline(nn.Module):
    """Hunyuan3DDiTFlowMatchingPipeline restated (batch 1)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.model = Hunyuan3DDiT(**cfg["dit"])
        self.vae = ShapeVAE(**cfg["vae"])
        self.conditioner = nn.Module()
        self.conditioner.main_image_encoder = DinoImageEncoder(cfg["cond"])
        self.eval()

    @torch.no_grad()
    def encode_cond(self, image):
        cond = self.conditioner.main_image_encoder(image)
        return torch.cat([cond, torch.zeros_like(cond)], dim=0)   # [cond, uncond]

    @torch.no_grad()
    def sample(self, cond2, latents, num_inference_steps, guidance_scale, trace=None, first_step=0, callback=None):
        """first_step / callback(i, latents): resume point and per-step hook of long runs (tools/make_cfg1_golden.py);
        the arithmetic of a step does not depend on them"""
        sig = flow_sigmas(num_inference_steps, self.cfg["sched"]["shift"])
        for i in range(first_step, num_inference_steps):
            t = torch.full((2,), float(sig[i]), dtype=latents.dtype)   # timesteps/num_train_timesteps == sigma
            v = self.model(torch.cat([latents] * 2), t, cond2)
            v_c, v_u = v.chunk(2)
            v = v_u + guidance_scale * (v_c - v_u)
            latents = latents + float(sig[i + 1] - sig[i]) * v
            if trace is not None:
                trace.append(latents.clone())
            if callback is not None:
                callback(i, latents)
        return latents

    @torch.no_grad()
    def latents_to_grid(self, latents, octree_resolution, num_chunks):
        z = self.vae(latents / self.vae.scale_factor)
        return volume_decode(self.vae, z, self.cfg["box_v"], octree_resolution, num_chunks), z

    @torch.no_grad()
    def __call__(self, image, num_inference_steps=50, octree_resolution=256, num_chunks=8000, generator=None,
                 guidance_scale=None):
        from . import mc
        g = self.cfg["guidance_scale"] if guidance_scale is None else guidance_scale
        img, _ = preprocess_image(image, **self.cfg["proc"])
        cond2 = self.encode_cond(img)
        lat = prepare_latents((1,) + self.vae.latent_shape, generator)
        lat = self.sample(cond2, lat, num_inference_steps, g)
        grid, _ = self.latents_to_grid(lat, octree_resolution, num_chunks)
        try:
            v, f = mc.hy3d_mesh(grid.numpy(), self.cfg["mc_level"], self.cfg["box_v"], octree_resolution)
        except (ValueError, RuntimeError):
            return None, grid
        return (v, f), grid


# ----------------------------------------------------------------------------- synthetic weights
def _is_norm_scale(k):
    return (k.endswith(".scale") or "norm" in k and k.endswith("weight") or k.endswith("ln_1.weight")
            or k.endswith("ln_2.weight") or k.endswith("ln_3.weight") or k.endswith("ln_post.weight")
            or "lambda1" in k or "layernorm.weight" in k)


def synthetic_state_dict(cfg, seed=0, std=0.02, mod_std=0.02, init="unit"):
    """Seeded synthetic weights with upstream key names ('model.', 'vae.', 'conditioner.' prefixes).

    init="unit" (default, the parity-grade checkpoint): every branch contributes O(1) to its residual stream, so a
    wiring error inside a block (QKV split order, concat order, shift/scale/gate chunk, GELU flavour, V layout ...)
    moves the output by far more than the bf16 tolerance (tests/test_mutation_cpu.py proves it hazard by hazard):
      Linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2), norm scales 1 +- 10 %, norm biases ~ N(0, 0.1^2),
      adaLN modulation layers ~ N(0, 9/fan_in) with bias N(0, 0.3^2) (|shift|, |scale|, |gate| ~ 0.3 .. 1),
      Dinov2 cls / position embeddings ~ N(0, 0.5^2).
    init="small": the round-1 checkpoint (Linear ~ N(0, std^2), near-identity blocks); kept for comparison only."""
    pipe = ShapePipeline(cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    if init == "unit":
        for prefix, mod in (("model.", pipe.model), ("vae.", pipe.vae), ("conditioner.", pipe.conditioner)):
            for k, p in mod.state_dict().items():
                if not torch.is_floating_point(p):
                    sd[prefix + k] = p.clone()
                    continue
                is_mod = ".lin." in k or "adaLN" in k
                if _is_norm_scale(k):
                    t = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
                elif k.endswith(("cls_token", "position_embeddings", "mask_token")):
                    t = 0.5 * torch.randn(p.shape, generator=g)
                elif p.ndim >= 2:
                    fan_in = p[0].numel()
                    t = (3.0 if is_mod else 1.0) / math.sqrt(fan_in) * torch.randn(p.shape, generator=g)
                else:
                    t = (0.3 if is_mod else 0.1) * torch.randn(p.shape, generator=g)
                sd[prefix + k] = t.to(torch.float32)
        return sd
    assert init == "small", init
    for prefix, mod in (("model.", pipe.model), ("vae.", pipe.vae), ("conditioner.", pipe.conditioner)):
        for k, p in mod.state_dict().items():
            if not torch.is_floating_point(p):
                sd[prefix + k] = p.clone()
                continue
            if k.endswith(".scale") or "norm" in k and k.endswith("weight") or k.endswith("ln_1.weight") \
                    or k.endswith("ln_2.weight") or k.endswith("ln_3.weight") or k.endswith("ln_post.weight") \
                    or "lambda1" in k or "layernorm.weight" in k:
                t = 1.0 + 0.05 * torch.randn(p.shape, generator=g)
            elif p.ndim >= 2:
                s = mod_std if (".lin." in k or "adaLN" in k) else std
                if "output_proj" in k:
                    s = 0.2
                t = s * torch.randn(p.shape, generator=g)
            else:
                t = 0.01 * torch.randn(p.shape, generator=g)
            sd[prefix + k] = t.to(torch.float32)
    return sd


def load_state_dict(pipe, sd):
    for prefix, mod in (("model.", pipe.model), ("vae.", pipe.vae), ("conditioner.", pipe.conditioner)):
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        mod.load_state_dict(sub, strict=True)
    return pipe

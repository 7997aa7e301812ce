"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the Stable-Diffusion-2.1-class UNet
blocks that upstream's texture stage is made of (reference call site
src/2d_to_3d_models/run.py:97 `pipeline_texgen(mesh, image=image)`, built at :126-128 / :207-209 -- upstream
hy3dgen/texgen runs a delighting UNet and a multiview UNet, both diffusers `UNet2DConditionModel`s on the SD-2.1 layout).

PARITY UNPINNED: neither hy3dgen nor diffusers is in the container and the reference holds no golden output for this
path.  What is restated is the published diffusers architecture (ResnetBlock2D, Transformer2DModel with
use_linear_projection, BasicTransformerBlock with GEGLU, Downsample2D, CrossAttnDownBlock2D, UNetMidBlock2DCrossAttn) under
diffusers' own module / parameter names, so that a real SD-2.1 `unet/diffusion_pytorch_model.safetensors` loads with
strict=True key for key.  tests/test_unet_cpu.py pins the building blocks against torch.nn.functional (group_norm, conv2d,
layer_norm, scaled_dot_product_attention) -- the primitives diffusers itself is built from.

Tensors are NCHW here (as in diffusers); the HIP path works on [H*W][C] rows (r3g/unet.py converts).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def sd21_config():
    """block_out_channels (320, 640, 1280, 1280), attention_head_dim (5, 10, 20, 20) = HEADS per block (head dim 64),
    cross_attention_dim 1024, layers_per_block 2, norm_num_groups 32 (stabilityai/stable-diffusion-2-1 unet/config.json)"""
    return dict(block_out_channels=(320, 640, 1280, 1280), heads=(5, 10, 20, 20), cross_attention_dim=1024, layers_per_block=2,
                groups=32, temb_dim=1280, ctx_tokens=77)


def small_config():
    """CI-sized: the same structure with 64 / 128 channels (head dim stays 64, groups 32 -> 2 / 4 channels per group)"""
    return dict(block_out_channels=(64, 128), heads=(1, 2), cross_attention_dim=128, layers_per_block=2, groups=32,
                temb_dim=256, ctx_tokens=13)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention: to_q / to_k / to_v without bias, to_out.0 with bias, scale 1/sqrt(64)"""

    def __init__(self, dim, heads, ctx_dim=None):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, L, C = x.shape
        d = C // self.heads
        q = self.to_q(x).view(B, L, self.heads, d).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], self.heads, d).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], self.heads, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
        return self.to_out[0]((w @ v).transpose(1, 2).reshape(B, L, C))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, ctx_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx):
        h = h + self.attn1(self.norm1(h))
        h = h + self.attn2(self.norm2(h), ctx)
        return h + self.ff(self.norm3(h))


class Transformer2DModel(nn.Module):
    """use_linear_projection=True (SD 2.x): GroupNorm(eps 1e-6) -> [B, HW, C] -> proj_in -> blocks -> proj_out -> + input"""

    def __init__(self, dim, heads, ctx_dim, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + x


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, heads, ctx_dim, layers=2, groups=32, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, groups) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x, temb, ctx):
        states = []
        for r, a in zip(self.resnets, self.attentions):
            x = a(r(x, temb), ctx)
            states.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            states.append(x)
        return x, states


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, temb_dim, heads, ctx_dim, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, groups), ResnetBlock2D(c, c, temb_dim, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, ctx_dim, groups)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        for a, r in zip(self.attentions, self.resnets[1:]):
            x = r(a(x, ctx), temb)
        return x


class DownBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, layers=2, groups=32, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x, temb, ctx=None):
        states = []
        for r in self.resnets:
            x = r(x, temb)
            states.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            states.append(x)
        return x, states


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpBlock2D(nn.Module):
    """diffusers get_up_block: resnet j takes cat(hidden, skip) with skip channels = in_channels for the last resnet,
    out_channels otherwise; hidden channels = prev_output_channel for the first resnet, out_channels otherwise"""

    def __init__(self, cin, cout, cprev, temb_dim, heads=None, ctx_dim=None, layers=3, groups=32, add_upsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D((cprev if j == 0 else cout) + (cin if j == layers - 1 else cout), cout,
                                                    temb_dim, groups) for j in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim, groups) for _ in range(layers)]) \
            if heads is not None else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, skips, temb, ctx=None):
        for j, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips[-1 - j]], dim=1), temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class UNet2DConditionModel(nn.Module):
    """diffusers UNet2DConditionModel on the SD-2.1 layout: CrossAttnDownBlock2D x (n-1) + DownBlock2D, mid block,
    UpBlock2D + CrossAttnUpBlock2D x (n-1); parameter names are diffusers' (conv_in, time_embedding.linear_1, down_blocks.i...,
    up_blocks.i.upsamplers.0.conv, conv_norm_out, conv_out)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        ch, heads, L, G = cfg["block_out_channels"], cfg["heads"], cfg["layers_per_block"], cfg["groups"]
        T, X = cfg["temb_dim"], cfg["cross_attention_dim"]
        n = len(ch)
        self.conv_in = nn.Conv2d(cfg.get("in_channels", 4), ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], T)
        downs = []
        for i in range(n):
            cin, cout, last = (ch[i - 1] if i else ch[0]), ch[i], i == n - 1
            downs.append(DownBlock2D(cin, cout, T, L, G, add_downsample=False) if last else
                         CrossAttnDownBlock2D(cin, cout, T, heads[i], X, L, G, add_downsample=True))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = UNetMidBlock2DCrossAttn(ch[-1], T, heads[-1], X, G)
        rch, rheads = list(reversed(ch)), list(reversed(heads))
        ups, prev = [], rch[0]
        for i in range(n):
            cout, cin = rch[i], rch[min(i + 1, n - 1)]
            ups.append(UpBlock2D(cin, cout, prev, T, None if i == 0 else rheads[i], X, L + 1, G, add_upsample=i < n - 1))
            prev = cout
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(G, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], cfg.get("out_channels", 4), 3, padding=1)

    def forward(self, sample, timestep, ctx, class_emb=None):
        """class_emb [B, temb_dim]: diffusers adds the class embedding to the time embedding (emb = emb + class_emb)"""
        emb = self.time_embedding(timestep_embedding(torch.as_tensor([float(timestep)]), self.cfg["block_out_channels"][0]))
        if class_emb is not None:
            emb = emb + class_emb
        x = self.conv_in(sample)
        skips = [x]
        for b in self.down_blocks:
            x, st = b(x, emb, ctx)
            skips += st
        x = self.mid_block(x, emb, ctx)
        for b in self.up_blocks:
            k = len(b.resnets)
            x = b(x, skips[-k:], emb, ctx)
            skips = skips[:-k]
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class UNetSlice(nn.Module):
    """first down block + the mid block of the UNet, under diffusers' names (down_blocks.0.*, mid_block.*): the slice the
    HIP path implements so far.  (conv_in, the other down / up blocks, conv_out: not on the HIP path yet.)"""

    def __init__(self, cfg):
        super().__init__()
        c0 = cfg["block_out_channels"][0]
        cm = cfg["block_out_channels"][-1]
        self.cfg = cfg
        self.down_blocks = nn.ModuleList([CrossAttnDownBlock2D(c0, c0, cfg["temb_dim"], cfg["heads"][0],
                                                               cfg["cross_attention_dim"], cfg["layers_per_block"], cfg["groups"])])
        self.mid_block = UNetMidBlock2DCrossAttn(cm, cfg["temb_dim"], cfg["heads"][-1], cfg["cross_attention_dim"], cfg["groups"])


def synthetic_state_dict(cfg, seed=0, full=False):
    """unit-scale weights: every branch moves its residual stream by O(0.3 .. 1) (as oracle/hy3d_torch.py's 'unit' init);
    full: the whole UNet2DConditionModel instead of the first-down-block + mid-block slice"""
    g = torch.Generator().manual_seed(seed)
    m = UNet2DConditionModel(cfg) if full else UNetSlice(cfg)
    sd = {}
    for k, v in m.state_dict().items():
        if v.ndim >= 2:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) / math.sqrt(fan_in)
        elif k.endswith("weight"):          # norm scales
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
    return sd


def load(cfg, sd, full=False):
    m = UNet2DConditionModel(cfg) if full else UNetSlice(cfg)
    m.load_state_dict(sd, strict=True)
    return m.eval()

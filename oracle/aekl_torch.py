"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the Stable-Diffusion-family VAE
(diffusers `AutoencoderKL`) that upstream's texture pipelines encode images with and decode latents through (SURVEY.md 8f
rank 3; reference call site src/2d_to_3d_models/run.py:97 `pipeline_texgen(mesh, image=image)`; both of upstream's
diffusion models -- the delighting InstructPix2Pix pipeline and the multiview pipeline -- carry an SD `vae/`).

PARITY UNPINNED: neither hy3dgen nor diffusers is in the container and the reference holds no golden output for this path.
What is restated is the published diffusers architecture (Encoder / Decoder of `autoencoder_kl`: ResnetBlock2D without a
time embedding, Downsample2D with padding 0 -- i.e. F.pad(x, (0, 1, 0, 1)) and a stride-2 convolution --, Upsample2D,
UNetMidBlock2D with ONE single-head attention whose head dim is the channel count, conv_norm_out + SiLU + conv_out, quant_conv /
post_quant_conv) under diffusers' own module / parameter names, so that a real `vae/diffusion_pytorch_model.safetensors` of
SD 1.x / 2.x loads with strict=True key for key: at SD dims the module has 83 653 863 parameters, the number diffusers reports
(tests/test_aekl_cpu.py).  The same file pins the blocks against torch.nn.functional.

Tensors are NCHW here (as in diffusers); the HIP path works on [H*W][C] rows (r3g/unet.py converts).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def sd_config():
    """stabilityai/stable-diffusion-2-1 vae/config.json: block_out_channels (128, 256, 512, 512), layers_per_block 2,
    latent_channels 4, norm_num_groups 32, in / out channels 3"""
    return dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, image_channels=3, groups=32)


def small_config():
    """CI-sized: three levels of 64 / 64 / 128 channels (groups 32 -> 2 / 4 channels per group)"""
    return dict(block_out_channels=(64, 64, 128), layers_per_block=1, latent_channels=4, image_channels=3, groups=32)


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1)"""

    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers Attention(channels, heads=1, dim_head=channels, bias=True, norm_num_groups, eps=1e-6,
    residual_connection=True, rescale_output_factor=1) as used by UNetMidBlock2D of the VAE"""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)                 # [b, hw, c]
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) * (float(c) ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(b, c, h, w)


class _Conv(nn.Module):
    """Downsample2D / Upsample2D keep their convolution in an attribute called `conv`"""

    def __init__(self, c, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=padding)


class Downsample2D(_Conv):
    def __init__(self, c):
        super().__init__(c, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0))


class Upsample2D(_Conv):
    def __init__(self, c):
        super().__init__(c, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Level(nn.Module):
    def __init__(self, resnets, sampler_name, sampler):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if sampler is not None:
            setattr(self, sampler_name, nn.ModuleList([sampler]))
        self._sampler_name = sampler_name if sampler is not None else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self._sampler_name:
            x = getattr(self, self._sampler_name)[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, L, g = cfg["block_out_channels"], cfg["layers_per_block"], cfg["groups"]
        self.conv_in = nn.Conv2d(cfg["image_channels"], ch[0], 3, padding=1)
        blocks, cin = [], ch[0]
        for i, c in enumerate(ch):
            res = [ResnetBlock2D(cin if j == 0 else c, c, g) for j in range(L)]
            blocks.append(_Level(res, "downsamplers", Downsample2D(c) if i < len(ch) - 1 else None))
            cin = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(ch[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, L, g = cfg["block_out_channels"], cfg["layers_per_block"], cfg["groups"]
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], g)
        blocks, cin = [], rev[0]
        for i, c in enumerate(rev):
            res = [ResnetBlock2D(cin if j == 0 else c, c, g) for j in range(L + 1)]
            blocks.append(_Level(res, "upsamplers", Upsample2D(c) if i < len(rev) - 1 else None))
            cin = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(g, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cfg["image_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = dict(cfg)
        z = cfg["latent_channels"]
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * z, 2 * z, 1)
        self.post_quant_conv = nn.Conv2d(z, z, 1)

    def encode_moments(self, x):
        """parameters of AutoencoderKL.encode(x).latent_dist: [b, 2 z, h/8, w/8] = (mean | logvar)"""
        return self.quant_conv(self.encoder(x))

    def encode_mode(self, x):
        """latent_dist.mode(): the mean (what InstructPix2Pix takes for its image latents)"""
        return self.encode_moments(x)[:, :self.cfg["latent_channels"]]

    def decode(self, z):
        """AutoencoderKL.decode(z).sample"""
        return self.decoder(self.post_quant_conv(z))


def build(cfg, seed=0):
    """random-init module with activations of order one through the depth (plain default init shrinks them layer by layer,
    which would hide errors of the deep layers from a relative comparison)"""
    g = torch.Generator().manual_seed(seed)
    m = AutoencoderKL(cfg)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m.eval()

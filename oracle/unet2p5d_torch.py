"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the multiview UNet of upstream's texture
stage -- [UPSTREAM-RECALLED] hy3dgen/texgen/hunyuanpaint/unet/modules.py `UNet2p5DConditionModel` / `Basic2p5DTransformerBlock`
(behind reference src/2d_to_3d_models/run.py:97; the model `Hunyuan3DPaintPipeline` generates its six views with).

PARITY UNPINNED, and more so than the plain UNet: hy3dgen is not in the container, so the structure below is what is recalled of
upstream's file, not a transcription --
  * the wrapped model is an SD-2.1 UNet2DConditionModel (oracle/unet_torch.py) whose conv_in takes 12 channels
    (noisy latent | normal-map latent | position-map latent);
  * every BasicTransformerBlock is wrapped: after `hidden = attn1(norm1(hidden)) + hidden`, and on the SAME norm1 output,
      - mode "w": the normalised states, views concatenated along the tokens ('(b n) l c -> b (n l) c'), are stored per block
        (`condition_embed_dict[layer_name]`),
      - mode "r" (and use_ra): hidden += ref_scale * attn_refview(norm_hidden, encoder_hidden_states = stored states),
      - more than one view (and use_ma): hidden += mva_scale * attn_multiview over the tokens of all views as one sequence;
    then attn2 (text) and the feed-forward as before; attn_multiview / attn_refview are diffusers Attention modules with the
    block's heads, no q/k/v bias, an output bias;
  * `class_embedding = nn.Embedding(cameras, temb_dim)` is added to the time embedding (class_labels = camera indices, offset by
    the number of reference slots for the generated views);
  * the reference pass runs a second copy of the ORIGINAL UNet (`unet_dual`, 4 input channels, plain blocks) at timestep 0 on the
    reference image's latents in mode "w"; the generation pass runs in mode "r" with the learned text embedding as context.
State-dict names follow that structure: "unet.<...>.transformer_blocks.0.transformer.<norm1|attn1|...>",
"unet.<...>.transformer_blocks.0.attn_multiview.to_q.weight", "unet.class_embedding.weight", "unet.learned_text_clip_gen",
"unet_dual.<plain names>".
"""
import math

import torch
import torch.nn as nn

from . import unet_torch as U


class Basic2p5DTransformerBlock(nn.Module):
    def __init__(self, transformer, layer_name, ctl, use_ma=True, use_ra=True):
        super().__init__()
        self.transformer = transformer
        self.layer_name = layer_name
        self.ctl = ctl                      # shared by all blocks of one UNet: what upstream passes as cross_attention_kwargs
        dim, heads = transformer.attn1.to_q.in_features, transformer.attn1.heads
        self.attn_multiview = U.Attention(dim, heads) if use_ma else None
        self.attn_refview = U.Attention(dim, heads) if use_ra else None

    def forward(self, h, ctx):
        t, c = self.transformer, self.ctl
        n = c.get("num_in_batch", 1)
        mode = c.get("mode", "")
        norm_h = t.norm1(h)
        h = h + t.attn1(norm_h)
        if "w" in mode:
            bn, l, ch = norm_h.shape
            c["condition_embed_dict"][self.layer_name] = norm_h.reshape(bn // n, n * l, ch)
        if "r" in mode and self.attn_refview is not None:
            cond = c["condition_embed_dict"][self.layer_name]                         # [b, n_ref l, c]
            cond = cond[:, None].expand(-1, n, -1, -1).reshape(-1, cond.shape[1], cond.shape[2])
            h = h + c.get("ref_scale", 1.0) * self.attn_refview(norm_h, cond)
        if n > 1 and self.attn_multiview is not None:
            bn, l, ch = norm_h.shape
            mv = norm_h.reshape(bn // n, n * l, ch)
            h = h + c.get("mva_scale", 1.0) * self.attn_multiview(mv).reshape(bn, l, ch)
        h = h + t.attn2(t.norm2(h), ctx)
        return h + t.ff(t.norm3(h))


def _wrap_blocks(unet, ctl, use_ma, use_ra):
    """replace every BasicTransformerBlock by its 2.5D wrapper; the layer name is the Transformer2DModel's prefix"""
    for name, mod in unet.named_modules():
        if isinstance(mod, U.Transformer2DModel):
            mod.transformer_blocks[0] = Basic2p5DTransformerBlock(mod.transformer_blocks[0], name, ctl, use_ma, use_ra)


class UNet2p5DConditionModel(nn.Module):
    max_num_ref_image = 5
    max_num_gen_image = 12 * 3 + 4 * 2

    def __init__(self, cfg):
        """cfg: oracle.unet_torch config of the wrapped SD-2.1 UNet (in_channels is overridden: 12 for the generator, 4 for
        the reference copy)"""
        super().__init__()
        self.cfg = dict(cfg)
        self.ctl, self.ctl_dual = {}, {}
        self.unet_dual = U.UNet2DConditionModel(dict(cfg, in_channels=4, out_channels=4))
        _wrap_blocks(self.unet_dual, self.ctl_dual, False, False)
        self.unet = U.UNet2DConditionModel(dict(cfg, in_channels=12, out_channels=4))
        _wrap_blocks(self.unet, self.ctl, True, True)
        self.unet.class_embedding = nn.Embedding(self.max_num_ref_image + self.max_num_gen_image, cfg["temb_dim"])
        self.unet.learned_text_clip_gen = nn.Parameter(torch.randn(1, cfg["ctx_tokens"], cfg["cross_attention_dim"]))
        self.unet.learned_text_clip_ref = nn.Parameter(torch.randn(1, cfg["ctx_tokens"], cfg["cross_attention_dim"]))
        # upstream gives the reference copy the same camera embedding (it is a deep copy made after the embedding exists)
        self.unet_dual.class_embedding = nn.Embedding(self.max_num_ref_image + self.max_num_gen_image, cfg["temb_dim"])

    @torch.no_grad()
    def reference_pass(self, ref_latents, camera_info_ref=None):
        """ref_latents [n_ref, 4, h, w] (one object) -> condition_embed_dict: layer name -> [1, n_ref h w, c]"""
        n = ref_latents.shape[0]
        cond = {}
        self.ctl_dual.clear()
        self.ctl_dual.update(mode="w", num_in_batch=n, condition_embed_dict=cond)
        ctx = self.unet.learned_text_clip_ref.expand(n, -1, -1)
        cls = self.unet_dual.class_embedding(camera_info_ref) if camera_info_ref is not None else None
        self.unet_dual(ref_latents, 0.0, ctx, class_emb=cls)
        return cond

    @torch.no_grad()
    def forward(self, sample, timestep, normal_imgs, position_imgs, cond, camera_info_gen=None, mva_scale=1.0, ref_scale=1.0,
                zero_context=False):
        """sample / normal_imgs / position_imgs [n_gen, 4, h, w] (one object: b = 1) -> noise prediction [n_gen, 4, h, w];
        zero_context: an all-zero text context instead of the learned embedding (the unconditional branch, see mvpaint_torch)"""
        n = sample.shape[0]
        x = torch.cat([sample, normal_imgs, position_imgs], dim=1)
        self.ctl.clear()
        self.ctl.update(mode="r", num_in_batch=n, condition_embed_dict=cond, mva_scale=mva_scale, ref_scale=ref_scale)
        ctx = self.unet.learned_text_clip_gen.expand(n, -1, -1)
        if zero_context:
            ctx = torch.zeros_like(ctx)
        cls = None
        if camera_info_gen is not None:
            cls = self.unet.class_embedding(camera_info_gen + self.max_num_ref_image)
        return self.unet(x, timestep, ctx, class_emb=cls)


def build(cfg, seed=0):
    """unit-scale random weights (as oracle.unet_torch.synthetic_state_dict): every branch moves its residual stream by O(1)"""
    g = torch.Generator().manual_seed(seed)
    m = UNet2p5DConditionModel(cfg)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "learned_text_clip" in name or "class_embedding" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if "class_embedding" in name else 1.0))
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m.eval()

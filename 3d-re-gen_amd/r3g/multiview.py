"""The multiview diffusion UNet of upstream's texture stage on the HIP blocks: [UPSTREAM-RECALLED]
hy3dgen/texgen/hunyuanpaint/unet/modules.py `UNet2p5DConditionModel` -- an SD-2.1 UNet whose transformer blocks also attend
across the views of the object (attn_multiview) and to the hidden states of a reference pass over the input image (attn_refview),
with a camera embedding added to the time embedding and 12 input channels (latent | normal-map latent | position-map latent).
Two library contexts: the generator and the plain 4-channel reference copy (`unet_dual`); the reference pass keeps norm1's output
of every transformer in HBM and the generator reads it from there (no copy, no host round trip).

This is the model only (one evaluation = r3g_unet_forward_mv over all views); upstream's sampling pipeline around it (VAE
encoding of the rendered normal / position maps, classifier-free guidance, scheduler) is not on this path yet."""
import torch

from . import unet as _unet


class MultiviewUNet:
    max_num_ref_image = 5       # upstream: camera indices of generated views are offset by the reference slots

    def __init__(self, state_dict, unet_config, n_views_max=6, n_ref_max=1, latent_hw=64 * 64, device=0):
        """state_dict: upstream's UNet2p5DConditionModel ("unet.*", "unet_dual.*"); unet_config: block_out_channels,
        layers_per_block, cross_attention_dim, ctx_tokens, temb_dim, groups"""
        gen_sd, ref_sd, extra = _unet.split_2p5d_state_dict(state_dict)
        c = unet_config
        ch = tuple(c["block_out_channels"])
        self.n_levels, self.layers = len(ch), c["layers_per_block"]
        common = dict(max_channels=2 * max(ch), temb_dim=c["temb_dim"], ctx_dim=c["cross_attention_dim"], groups=c["groups"],
                      device=device, block_out_channels=ch, layers_per_block=c["layers_per_block"], out_channels=4)
        self.gen = _unet.UnetBlocks(gen_sd, max_hw=n_views_max * latent_hw, ctx_tokens=max(c["ctx_tokens"], n_ref_max * latent_hw),
                                    in_channels=12, **common)
        self.ref = _unet.UnetBlocks(ref_sd, max_hw=n_ref_max * latent_hw, ctx_tokens=c["ctx_tokens"], in_channels=4, **common)
        self.text_gen = extra["learned_text_clip_gen"].detach().reshape(1, -1, c["cross_attention_dim"])
        self.text_ref = extra["learned_text_clip_ref"].detach().reshape(1, -1, c["cross_attention_dim"])
        self.has_reference = False

    def reference_pass(self, ref_latents, camera_info_ref=None):
        """ref_latents NCHW [n_ref, 4, h, w] of ONE object: the reference copy runs once at timestep 0 and keeps every
        transformer's normalised hidden states; the generator is pointed at them"""
        self.ref.forward_mv(ref_latents, 0.0, self.text_ref, class_labels=camera_info_ref, flags=1)
        for p in _unet.transformer_prefixes(self.n_levels, self.layers):
            self.gen.set_condition(p, self.ref)
        self.has_reference = True

    def __call__(self, sample, timestep, normal_imgs, position_imgs, camera_info_gen=None, mva_scale=1.0, ref_scale=1.0):
        """sample / normal_imgs / position_imgs NCHW [n_views, 4, h, w] -> noise prediction [n_views, 4, h, w]"""
        x = torch.cat([sample, normal_imgs, position_imgs], dim=1)
        labels = None if camera_info_gen is None else [int(v) + self.max_num_ref_image for v in camera_info_gen]
        return self.gen.forward_mv(x, timestep, self.text_gen, class_labels=labels, flags=2 if self.has_reference else 0,
                                   mva_scale=mva_scale, ref_scale=ref_scale)

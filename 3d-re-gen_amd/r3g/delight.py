"""The delighting model of upstream's texture stage as one object: UNet (8 input channels: latent | image latent), SD VAE and
Euler-ancestral sampling loop, all on the HIP blocks.  [UPSTREAM-RECALLED] hy3dgen/texgen/utils/dehighlight_utils.py builds a
diffusers StableDiffusionInstructPix2PixPipeline, swaps its scheduler for EulerAncestralDiscreteScheduler and calls it with
prompt "", guidance_scale 1.0 and image_guidance_scale 1.5 -- with guidance_scale 1.0 diffusers runs NO classifier-free
guidance (`guidance_scale > 1.0 and image_guidance_scale >= 1.0` is false): every step is one UNet evaluation on
cat(latents / sqrt(sigma^2 + 1), image_latents).  The prompt's text embedding is an input here (for the fixed prompt "" it is
one constant [77, 1024] tensor; the CLIP text encoder that produces it is not part of this path).

The loop keeps everything in HBM as rows: latents f32 [h*w][4], UNet input f32 [h*w][8], noise rows uploaded once."""
import torch

from . import sched as _sched
from . import unet as _unet


class InstructPix2Pix:
    def __init__(self, unet_state, vae_state, unet_config, vae_config, image_size=512, scaling_factor=0.18215,
                 prediction_type="epsilon", device=0):
        """unet_config: block_out_channels, layers_per_block, cross_attention_dim, ctx_tokens, temb_dim, groups (diffusers'
        unet/config.json names where they exist); vae_config: block_out_channels, layers_per_block, latent_channels,
        image_channels, groups.  image_size: the largest square image the instance is sized for."""
        ch = tuple(unet_config["block_out_channels"])
        self.vae = _unet.AutoencoderKLBlocks(vae_state, block_out_channels=tuple(vae_config["block_out_channels"]),
                                             layers_per_block=vae_config["layers_per_block"],
                                             latent_channels=vae_config["latent_channels"],
                                             image_channels=vae_config["image_channels"], groups=vae_config["groups"],
                                             max_image_hw=image_size * image_size, device=device)
        zc = self.vae.latent_channels
        lat = image_size // self.vae.factor
        self.unet = _unet.UnetBlocks(unet_state, max_hw=lat * lat, max_channels=2 * max(ch), temb_dim=unet_config["temb_dim"],
                                     ctx_dim=unet_config["cross_attention_dim"], ctx_tokens=unet_config["ctx_tokens"],
                                     groups=unet_config["groups"], device=device, block_out_channels=ch,
                                     layers_per_block=unet_config["layers_per_block"], in_channels=2 * zc, out_channels=zc)
        self.scheduler = _sched.EulerAncestralDiscrete(prediction_type=prediction_type)
        self.scaling_factor = float(scaling_factor)
        self.device = self.vae.device

    def close(self):
        """give the two contexts' HBM back"""
        self.unet.close()
        self.vae.close()

    def __call__(self, image, prompt_embeds, num_inference_steps=50, generator=None, latents=None, step_noise=None,
                 output="image"):
        """image NCHW [1, 3, H, W] in [-1, 1]; prompt_embeds [1, tokens, ctx_dim].  Noise: `latents` [1, z, h, w] and
        `step_noise` (one [1, z, h, w] per step) when given, else drawn from `generator` (a CPU torch.Generator, as upstream's
        torch.manual_seed(42): first the initial latents, then one draw per step, fp32).  -> NCHW [1, 3, H, W] on the device (output="latent": the final latents NCHW [1, z, h, w], before the
        division by the scaling factor and the decoder)"""
        _, _, H, W = image.shape
        f, zc = self.vae.factor, self.vae.latent_channels
        h, w = H // f, W // f
        n = int(num_inference_steps)
        if latents is None:
            latents = torch.randn((1, zc, h, w), generator=generator, dtype=torch.float32)
        if step_noise is None:
            step_noise = [torch.randn((1, zc, h, w), generator=generator, dtype=torch.float32) for _ in range(n)]
        if len(step_noise) != n:
            raise ValueError("step_noise must hold one draw per step")
        sch = self.scheduler.set_timesteps(n)
        dev = self.device
        ctx = prompt_embeds[0].to(dev, torch.bfloat16).contiguous()
        image_latents = self.vae.encode(image)[:, :zc]                              # latent_dist.mode(), not scaled
        il_rows = _unet.to_rows(image_latents)
        x = _unet.to_rows(latents.to(dev)) * sch.init_noise_sigma
        noise = [_unet.to_rows(e.to(dev)) for e in step_noise]
        inp = torch.empty((h * w, 2 * zc), dtype=torch.float32, device=dev)
        eps = torch.empty((h * w, zc), dtype=torch.float32, device=dev)
        for i in range(n):
            sch.model_input(x, il_rows, i, out=inp)
            self.unet.forward_rows(inp, h, w, float(sch.timesteps[i]), ctx, out=eps)
            sch.step(x, eps, noise[i], i)
        if output == "latent":
            return _unet.from_rows(x, h, w)
        rows = self.vae.decode_rows(x / self.scaling_factor, h, w)
        return _unet.from_rows(rows[:, :self.vae.image_channels], H, W)

"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the sampling loop of upstream's multiview
texture model -- [UPSTREAM-RECALLED] hy3dgen/texgen/hunyuanpaint/pipeline.py `HunyuanPaintPipeline` as
hy3dgen/texgen/utils/multiview_utils.py `Multiview_Diffusion_Net` calls it (30 steps, EulerAncestralDiscreteScheduler with
timestep_spacing "trailing", view size 512, guidance_scale 2.0, camera_info_ref [[0]]).  Behind reference
src/2d_to_3d_models/run.py:97.  PARITY UNPINNED (no hy3dgen, no diffusers, no checkpoint in the container): the loop below is
what is recalled -- VAE-encode the reference image, the normal maps and the position maps (latent_dist.sample() * scaling factor);
one reference pass; per step cat(latents / sqrt(sigma^2 + 1), normal latents, position latents) through the 2.5D UNet with the
reference attention on and, as the unconditional branch of classifier-free guidance, with it off; Euler-ancestral update; decode."""
import numpy as np
import torch

from . import pix2pix_torch as P


def trailing_tables(num_inference_steps, num_train_timesteps=1000):
    ts = np.round(np.arange(num_train_timesteps, 0, -num_train_timesteps / num_inference_steps)) - 1
    s = np.interp(ts, np.arange(num_train_timesteps), P.train_sigmas(num_train_timesteps))
    return ts.astype(np.float32), np.concatenate([s, [0.0]]).astype(np.float32)


def encode(vae, images, noise, scaling_factor):
    zc = vae.cfg["latent_channels"]
    mom = vae.encode_moments(images)
    mean, logvar = mom[:, :zc], mom[:, zc:].clamp(-30.0, 20.0)
    return (mean + torch.exp(0.5 * logvar) * noise) * scaling_factor


@torch.no_grad()
def multiview_paint(unet, vae, ref_images, normal_imgs, position_imgs, camera_info_gen, camera_info_ref, num_inference_steps, noise,
                    guidance_scale=2.0, scaling_factor=0.18215, output="image", uncond_context="zeros"):
    """unet: oracle.unet2p5d_torch.UNet2p5DConditionModel; vae: oracle.aekl_torch.AutoencoderKL; noise as r3g.multiview.
    uncond_context "zeros": [UPSTREAM-RECALLED] negative_prompt_embeds = zeros_like(prompt_embeds) in the unconditional branch;
    "learned": the learned embedding in both branches (what rounds 2-3 restated)"""
    ts, sig = trailing_tables(num_inference_steps)
    ref_latents = encode(vae, ref_images, noise["ref"], scaling_factor)
    nl = encode(vae, normal_imgs, noise["normal"], scaling_factor)
    pl = encode(vae, position_imgs, noise["position"], scaling_factor)
    cond = unet.reference_pass(ref_latents, None if camera_info_ref is None else torch.as_tensor(camera_info_ref))
    cam = None if camera_info_gen is None else torch.as_tensor(camera_info_gen)
    x = noise["latents"] * float(sig.max())
    for i, t in enumerate(ts):
        xs = x / (float(sig[i]) ** 2 + 1) ** 0.5
        eps = unet(xs, float(t), nl, pl, cond, cam)
        if guidance_scale > 1.0:
            eps_u = unet(xs, float(t), nl, pl, cond, cam, ref_scale=0.0, zero_context=uncond_context == "zeros")
            eps = eps_u + guidance_scale * (eps - eps_u)
        x = P.euler_ancestral_step(x, eps, noise["steps"][i], sig[i], sig[i + 1])
    if output == "latent":
        return x
    if output == "both":           # (final latents, decoded views) of ONE run of the loop
        return x, vae.decode(x / scaling_factor)
    return vae.decode(x / scaling_factor)
